"""Build libdepgan.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU; the shared object stays next to this file so
it travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import fcntl
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdepgan.so")
SOURCES = ["igemm_conv.hip", "igemm_wp.hip", "igemm_wino.hip", "igemm_bf16.hip", "deconv_fwd.hip", "deconv_wgrad.hip", "wgrad.hip", "wgrad_bf16.hip", "direct.hip", "ops.hip", "noise.hip", "train_ops.hip", "model.hip",
           "uresnet.hip", "data.hip"]
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def source_hash():
    """sha256 over every source the library is built from (csrc/*.hip, *.h, *.inc and include/depgan.h), in name order.
    It is compiled into the library (depgan_source_hash()), so a stale libdepgan.so -- e.g. one cross-compiled from
    other sources and shipped next to these -- is detected at load time instead of silently running old kernels."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc")))
    files.append(os.path.join(HERE, "..", "include", "depgan.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:32]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile what is stale and link.  Safe to call from several processes at once (one rank per GPU under
    torch.distributed.run): an exclusive file lock serialises them, objects and the library are replaced atomically."""
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    with open(os.path.join(HERE, "build", ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "depgan.h"))
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    flags = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value"]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])
    # the source hash lives in a translation unit of its own, rebuilt whenever it changes (i.e. whenever anything does)
    digest = source_hash()
    stamp_src = os.path.join(objdir, "srchash.cpp")
    stamp_obj = os.path.join(objdir, "srchash.o")
    text = 'extern "C" const char* depgan_source_hash(void) { return "%s"; }\n' % digest
    if not os.path.exists(stamp_src) or open(stamp_src).read() != text:
        with open(stamp_src, "w") as fh:
            fh.write(text)
    if force or _stale(stamp_obj, [stamp_src]):
        jobs.append(["g++", "-O1", "-fPIC", "-c", stamp_src, "-o", stamp_obj])

    def run(cmd):
        # the output is written next to its final name and renamed, so a reader never sees a partial file
        out = cmd[-1] if cmd[-2] == "-o" else cmd[cmd.index("-o") + 1]
        tmp = out + ".tmp%d" % os.getpid()
        real = [tmp if c == out else c for c in cmd]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(real, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        os.replace(tmp, out)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES] + [stamp_obj]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
