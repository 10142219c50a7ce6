"""Hash of the HIP library's sources (no dependency on the library itself, so __graft_entry__.build() can load this file
on its own before a stale libmmsum_hip.so is replaced)."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_build_id():
    """The id csrc/Makefile bakes into the library: first 16 hex digits of the SHA-256 over its sources (HASH_SRCS order)."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    names = sorted(os.path.basename(f) for pat in ("*.hip", "*.h", "*.inc") for f in glob.glob(os.path.join(csrc, pat)))
    files = [os.path.join(csrc, n) for n in names if n != "build_id.inc"]
    files += [os.path.join(_HERE, "..", "include", "mmsum_hip.h"), os.path.join(csrc, "Makefile"), os.path.join(csrc, "check_resources.py"),
              os.path.join(csrc, "check_handoff.py")]
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
