"""Model shape contract: the fields of /root/reference/cfg/bart-large.json that the hot path reads
(via /root/reference/src/transformer/configuration_bart.py:36-128)."""
import json


class BartConfig:
    def __init__(self, vocab_size=50265, d_model=1024, encoder_ffn_dim=4096, decoder_ffn_dim=4096, encoder_layers=12,
                 decoder_layers=12, encoder_attention_heads=16, decoder_attention_heads=16, max_position_embeddings=1024,
                 dropout=0.1, attention_dropout=0.0, activation_dropout=0.0, activation_function="gelu", init_std=0.02,
                 pad_token_id=1, bos_token_id=0, eos_token_id=2, decoder_start_token_id=2, extra_pos_embeddings=2,
                 normalize_before=False, normalize_embedding=True, scale_embedding=False,
                 static_position_embeddings=False, add_final_layer_norm=False, max_length=20, min_length=0, num_beams=1,
                 early_stopping=False, length_penalty=1.0, no_repeat_ngram_size=0, **unused):
        self.vocab_size = vocab_size
        self.d_model = d_model
        self.encoder_ffn_dim = encoder_ffn_dim
        self.decoder_ffn_dim = decoder_ffn_dim
        self.encoder_layers = encoder_layers
        self.decoder_layers = decoder_layers
        self.encoder_attention_heads = encoder_attention_heads
        self.decoder_attention_heads = decoder_attention_heads
        self.max_position_embeddings = max_position_embeddings
        self.dropout = dropout
        self.attention_dropout = attention_dropout
        self.activation_dropout = activation_dropout
        self.activation_function = activation_function
        self.init_std = init_std
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.decoder_start_token_id = decoder_start_token_id
        self.extra_pos_embeddings = extra_pos_embeddings
        self.normalize_before = normalize_before
        self.normalize_embedding = normalize_embedding
        self.scale_embedding = scale_embedding
        self.static_position_embeddings = static_position_embeddings
        self.add_final_layer_norm = add_final_layer_norm
        # generation defaults (transformers PretrainedConfig defaults; generate() arguments override them)
        self.max_length, self.min_length, self.num_beams = max_length, min_length, num_beams
        self.early_stopping, self.length_penalty, self.no_repeat_ngram_size = early_stopping, length_penalty, no_repeat_ngram_size
        self.validate()

    def validate(self):
        """The HIP path implements the BART-large family configuration the reference trains
        (post-LN, learned positions, LN on embeddings, GELU, no attention/activation dropout)."""
        assert self.d_model % 256 == 0 and self.d_model // self.encoder_attention_heads == 64, \
            "HIP path: d_model must be a multiple of 256 with head_dim 64"
        assert self.encoder_attention_heads == self.decoder_attention_heads
        assert not self.normalize_before and self.normalize_embedding and not self.scale_embedding
        assert not self.static_position_embeddings and not self.add_final_layer_norm
        assert self.activation_function == "gelu" and self.attention_dropout == 0.0 and self.activation_dropout == 0.0
        assert self.encoder_ffn_dim % 64 == 0 and self.decoder_ffn_dim % 64 == 0

    @property
    def heads(self):
        return self.encoder_attention_heads

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls(**json.load(f))

    def to_dict(self):
        return dict(self.__dict__)
