#!/usr/bin/env python3
"""Builds libyolo4hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU).

  python build.py            incremental: a translation unit is recompiled when the CONTENT of its source, of any header
                             of this directory / include/yolo4hip.h, the flags or the compiler version changed (a sha256
                             stamp next to each object -- mtimes are not trusted: a fresh checkout or a pushed snapshot
                             resets them)
  Y4_CLEAN=1 python build.py removes build/ and the .so first (what a clean tree does anyway: objects are git-ignored)
  Y4_JOBS=N                  parallel hipcc processes (default: min(8, cpus))
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "yolo4hip", "libyolo4hip.so")
BUILD = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]
# biggest translation units first so that the parallel build ends as early as possible
UNITS = ["conv_igemm_bf16", "conv_igemm_f16", "conv_p8_bf16", "conv_p8_f16", "conv_halo_bf16", "conv_halo_f16", "conv_halo2_bf16", "conv_halo2_f16", "conv_igemm_bf16_fused", "conv_igemm_f16_fused", "conv_igemm_f32",
         "resblock", "csp_stage", "stem_down", "misc_kernels", "decode_nms", "runtime", "conv_igemm"]


def sha(paths, extra):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode()); h.update(b"\0")      # (the NAME, not the path: a pushed snapshot lives elsewhere and must not rebuild)
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(extra.encode())
    return h.hexdigest()


def main():
    if os.environ.get("Y4_CLEAN") == "1":
        shutil.rmtree(BUILD, ignore_errors=True)
        if os.path.exists(OUT):
            os.remove(OUT)
    os.makedirs(BUILD, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    version = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    headers = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h"))
    headers.append(os.path.normpath(os.path.join(HERE, "..", "..", "include", "yolo4hip.h")))
    units = [u for u in UNITS if os.path.exists(os.path.join(HERE, u + ".hip"))]
    todo = []
    for u in units:
        src, obj, stamp = os.path.join(HERE, u + ".hip"), os.path.join(BUILD, u + ".o"), os.path.join(BUILD, u + ".sha")
        want = sha([src] + headers, " ".join(FLAGS) + version)
        have = open(stamp).read().strip() if os.path.exists(stamp) and os.path.exists(obj) else ""
        if have != want:
            todo.append((u, src, obj, stamp, want))

    def compile_one(job):
        u, src, obj, stamp, want = job
        r = subprocess.run([hipcc] + FLAGS + ["-c", src, "-o", obj], cwd=HERE, capture_output=True, text=True)
        if r.returncode != 0:
            return u, r.stderr[-6000:]
        if r.stderr.strip():
            sys.stderr.write(r.stderr[-3000:])
        with open(stamp, "w") as f:
            f.write(want)
        return u, None

    jobs = int(os.environ.get("Y4_JOBS", min(8, os.cpu_count() or 1)))
    failed = False
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        for u, err in ex.map(compile_one, todo):
            if err:
                failed = True
                sys.stderr.write(f"compile of {u}.hip failed:\n{err}\n")
    if failed:
        sys.exit(1)
    objs = [os.path.join(BUILD, u + ".o") for u in units]
    if todo or not os.path.exists(OUT):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs, check=True)
    # every kernel must have its host stub (a target builtin inside a template can silently drop it)
    und = subprocess.run(["nm", "-D", "--undefined-only", OUT], capture_output=True, text=True).stdout
    bad = [l for l in und.splitlines() if "_ZN2y4" in l]
    if bad:
        sys.stderr.write("error: undefined y4 symbols in %s:\n%s\n" % (OUT, "\n".join(bad[:5])))
        sys.exit(1)
    print(f"built {os.path.normpath(OUT)} ({len(todo)} of {len(units)} translation units recompiled)")


if __name__ == "__main__":
    main()
