"""Register / LDS / scratch budget of every kernel in libp2e_hip.so, read from the code objects' metadata (no GPU
needed).  Usage: python tools/kernel_resources.py [lib.so] [substring ...]   ->  name, VGPRs, AGPRs, SGPRs, LDS bytes,
scratch bytes, waves per SIMD the VGPR budget allows (512 / VGPRs, at most 8)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(ROOT, "plonky2-ecdsa_amd", "libp2e_hip.so")
filt = [a for a in sys.argv[1:] if not a.endswith(".so")]
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
with tempfile.TemporaryDirectory() as td:
    import shutil
    copy = os.path.join(td, "lib.so")          # llvm-objdump writes the bundles next to the file it reads
    shutil.copy(lib, copy)
    subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", copy], cwd=td, text=True, stderr=subprocess.STDOUT)
    objs = [os.path.join(td, f) for f in os.listdir(td) if "gfx950" in f]
    rows = {}
    for o in objs:
        notes = subprocess.check_output([READELF, "--notes", o], text=True)
        for blk in notes.split("- .agpr_count:")[1:]:
            blk = ".agpr_count:" + blk
            g = lambda key: re.search(r"\." + key + r":\s*(\S+)", blk)
            name = g("name").group(1)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            dem = re.sub(r"\(.*", "", dem).replace("void ", "")
            rows[dem] = (int(g("vgpr_count").group(1)), int(g("agpr_count").group(1)), int(g("sgpr_count").group(1)),
                         int(g("group_segment_fixed_size").group(1)), int(g("private_segment_fixed_size").group(1)))
print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'LDS':>7s} {'scratch':>8s} {'waves/SIMD':>10s}")
for name in sorted(rows):
    if filt and not any(f in name for f in filt):
        continue
    v, a, s, l, p = rows[name]
    tot = max(1, v + a)
    print(f"{name:60s} {v:5d} {a:5d} {s:5d} {l:7d} {p:8d} {min(8, 512 // ((tot + 7) // 8 * 8)):10d}")
