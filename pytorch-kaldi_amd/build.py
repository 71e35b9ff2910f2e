"""Build libpk_amd.so (the C-ABI HIP library) in-tree for gfx950.

    python -m pytorch-kaldi_amd.build      (or __graft_entry__.build())

hipcc cross-compiles without a GPU.  Objects are cached by source mtime under
``pytorch-kaldi_amd/lib/obj`` so that an unchanged tree rebuilds in seconds.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libpk_amd.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def _newest_header():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "..", "include", "pk_amd.h"))
    return max(os.path.getmtime(d) for d in deps)


def _compile(hipcc, src, obj):
    cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stdout))
    return r.stdout


def build(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link lib/libpk_amd.so.  Returns the path."""
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdr_t = _newest_header()
    jobs, objs = [], []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJDIR, f[:-4] + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t)
        if stale:
            jobs.append((src, obj))
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            futs = {ex.submit(_compile, hipcc, s, o): s for s, o in jobs}
            for fut in concurrent.futures.as_completed(futs):
                out = fut.result()
                if verbose:
                    print("compiled", os.path.basename(futs[fut]), out.strip())
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", LIB]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
