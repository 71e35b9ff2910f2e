"""Per-kernel resource usage of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage, cross-compiled):
    python tools/kres.py pytorch-kaldi_amd/csrc/pk_rec_persist3.hip [name-filter]
Columns: VGPRs, AGPRs, SGPRs, VGPR spills, SGPR spills, scratch bytes, static LDS, waves / SIMD."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark: +(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    txt = m.group(1).strip()
    if txt.startswith("Function Name:") or txt.startswith("Name:"):
        cur = txt.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in txt:
        k, v = txt.split(":", 1)
        rows[cur][k.strip()] = v.strip()
print("%-100s %5s %5s %5s %6s %6s %7s %6s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "vspill", "sspill", "scratch", "LDS", "occ"))
for name, r in rows.items():
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
    except OSError:
        dem = name
    dem = dem.replace("(anonymous namespace)::", "")
    if flt and flt not in dem:
        continue
    print("%-100s %5s %5s %5s %6s %6s %7s %6s %4s" % (dem[:100], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs", r.get("SGPRs")),
                                                   r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
                                                   r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
