"""sha256 over the kernel sources and their Makefile (file name + bytes, sorted by name): what miso_version() embeds, so that a
stale libmiso_hip.so cannot pass for the tree it travels with (tests/test_capi_symbols.py compares)."""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def source_files():
    files = [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith((".hip", ".hpp", ".inc"))
             and f != "version.inc"]
    files.append(os.path.join(HERE, "Makefile"))      # (per-file compiler flags are part of what the library is)
    files.append(os.path.normpath(os.path.join(HERE, "..", "..", "include", "miso_hip.h")))
    return files


def source_hash() -> str:
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    out = f'#define MISO_SOURCE_HASH "{source_hash()}"\n'
    path = os.path.join(HERE, "version.inc")
    if len(sys.argv) > 1 and sys.argv[1] == "--write":
        if not os.path.exists(path) or open(path).read() != out:
            open(path, "w").write(out)
    else:
        print(source_hash())
