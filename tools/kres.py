"""dev: per-kernel register / LDS / spill counts of a compiled HIP object.
   python tools/kres.py miso_amd/csrc/grad_pull.o [regex]"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    obj, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else ".")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, tmp + "/fb.bin"])
        subprocess.check_call([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + tmp + "/fb.bin",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + tmp + "/k.co"])
        txt = subprocess.check_output([LLVM + "llvm-readelf", "--notes", tmp + "/k.co"], text=True)
    rows = []
    for blk in txt.split(".agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        rows.append((g("name"), g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
                     g("group_segment_fixed_size"), g("private_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), text=True, capture_output=True).stdout.split("\n")
    for r, nm in zip(rows, names):
        nm = re.sub(r"\(.*", "", nm)
        if re.search(pat, nm):
            print(f"vgpr {r[1]:>4} spill {r[2]:>3} sgpr {r[3]:>4} sspill {r[4]:>3} lds {r[5]:>6} scratch {r[6]:>5}  {nm}")


if __name__ == "__main__":
    main()
