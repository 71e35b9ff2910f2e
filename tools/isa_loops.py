"""Per-loop census of a HIP source's gfx950 ISA: scratch loads / stores, MFMAs and workgroup barriers inside every
depth-1 loop that contains MFMAs (the time loops of the persistent recurrences).  Used for DESIGN.md 6.1: a
`scratch_load` inside a time loop is a register the allocator gave up on, and one that is followed by
`s_waitcnt vmcnt(0)` while global loads are in flight puts HBM latency on the step's dependency chain.

    python tools/isa_loops.py pytorch-kaldi_amd/csrc/pk_rec_persist2_lstm.hip [kernel-name-substring]

Needs hipcc only (cross-compiles without a GPU)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def census(asm_text, want=""):
    kern, cur, stats = None, None, {}
    for line in asm_text.split("\n"):
        m = re.match(r"^(_Z\S+):\s", line + " ")
        if m and "kernel" in m.group(1):
            kern, cur = m.group(1), None
            continue
        if kern is None:
            continue
        if line.startswith(".Lfunc_end"):
            kern = None
            continue
        m = re.match(r"^\.LBB\d+_\d+:\s*(;.*)?$", line)
        if m:
            c = m.group(1) or ""
            h = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", c)
            if "This Loop Header: Depth=1" in c:
                cur = line.split(":")[0][2:]
            elif h:
                if h.group(2) == "1":
                    cur = h.group(1)
            elif "Parent Loop" not in c and "Inner Loop" not in c:
                cur = None
            continue
        op = line.strip().split()[0] if line.strip() else ""
        d = stats.setdefault((kern, cur), {"scratch_load": 0, "scratch_store": 0, "mfma": 0, "barrier": 0, "instructions": 0})
        d["instructions"] += 1
        if op.startswith("scratch_load"):
            d["scratch_load"] += 1
        elif op.startswith("scratch_store"):
            d["scratch_store"] += 1
        elif op.startswith("v_mfma"):
            d["mfma"] += 1
        elif op == "s_barrier":
            d["barrier"] += 1
    return {k: v for k, v in stats.items() if k[1] and v["mfma"] > 0 and want in k[0]}


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out]
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        text = open(out).read()
    for (kern, loop), d in census(text, want).items():
        demangled = subprocess.run(["c++filt", kern], stdout=subprocess.PIPE, text=True).stdout.strip() or kern
        print("%-90s loop %-10s %s" % (demangled[:90], loop, d))


if __name__ == "__main__":
    main()
