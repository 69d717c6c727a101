"""Hash of the kernel sources of this tree (the value miso_version() of a fresh build embeds)."""
import importlib.util
import os

_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "srchash.py")
_spec = importlib.util.spec_from_file_location("miso_amd._srchash", _path)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
source_hash = _mod.source_hash
