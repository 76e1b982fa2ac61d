"""Build csrc/*.hip into csrc/libwsi_hgnn.so for gfx950 (hipcc cross-compiles without a GPU).

The shared object is built IN-TREE (it is git-ignored but travels to the GPU box with the
repo snapshot).  ``python -m wsi_hgnn_amd.build`` or ``__graft_entry__.build()``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libwsi_hgnn.so")
LIB_ABLATE = os.path.join(CSRC, "libwsi_hgnn_ablate.so")      # measurement build (-DWSI_ABLATE): tools/ only, never loaded by the package itself
SOURCES = ["error.hip", "heat_attn.hip", "heat_attn_tiled.hip", "gemm_f32.hip", "gemm_emu16.hip", "gemm_tn16.hip", "segment.hip", "rowwise.hip", "knn.hip", "asap.hip", "optim.hip", "loss.hip", "pooled.hip", "plan.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
         "-Wno-unused-result"]


def _stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "wsi_hgnn.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_native(force: bool = False, verbose: bool = True, ablate: bool = False) -> str:
    """``ablate=True`` builds the measurement flavour (kernel variants and environment knobs compiled in, csrc/common.h::knob) next to
    the product library; the product library contains neither.  Every source is compiled to an object of its own (in parallel, only when it or a
    header changed) and the objects are linked into the shared library."""
    lib = LIB_ABLATE if ablate else LIB
    if not force and not _stale(lib):
        return lib
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    # several ranks may import at once (torchrun): serialise the build, re-check staleness under the lock
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale(lib):
                return lib
            hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
            if not os.path.exists(hipcc):
                raise RuntimeError("hipcc not found: cannot build libwsi_hgnn.so")
            objdir = os.path.join(CSRC, ".obj_ablate" if ablate else ".obj")
            os.makedirs(objdir, exist_ok=True)
            headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
            headers.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "wsi_hgnn.h"))
            hdr_time = max(os.path.getmtime(h) for h in headers if os.path.exists(h))
            cflags = [f for f in FLAGS if f != "-shared"] + (["-DWSI_ABLATE"] if ablate else [])

            def compile_one(src: str) -> str:
                obj = os.path.join(objdir, src.replace(".hip", ".o"))
                path = os.path.join(CSRC, src)
                if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), hdr_time):
                    cmd = [hipcc] + cflags + ["-c", path, "-o", obj]
                    if verbose:
                        print("[wsi_hgnn_amd.build]", " ".join(cmd), file=sys.stderr, flush=True)   # stderr: bench.py's stdout is one JSON line
                    subprocess.run(cmd, check=True)
                return obj

            with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
                objs = list(pool.map(compile_one, SOURCES))
            tmp = f"{lib}.{os.getpid()}.tmp"
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", tmp]
            if verbose:
                print("[wsi_hgnn_amd.build]", " ".join(cmd), file=sys.stderr, flush=True)
            subprocess.run(cmd, check=True)
            os.replace(tmp, lib)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, ablate="--ablate" in sys.argv))
