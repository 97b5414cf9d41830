"""Build the C-ABI shared library (gfx950 code objects) in-tree with hipcc.

    python -m brdf_nerf_amd.build [--verbose]

Produces brdf_nerf_amd/libbrdfnerf_hip.so.  hipcc cross-compiles without a GPU.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbrdfnerf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-Wno-pass-failed", "-Wno-unused-result"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def source_hash():
    """sha256 (16 hex digits) over everything the library is compiled from: csrc/, the C ABI header and this file (the per-file
    compiler flags).  Compiled into the library (bn_source_hash); also the key of the committed PMC passes (bench.py)."""
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")))
    for f in files + [os.path.join(ROOT, "include", "brdfnerf_hip.h"), os.path.abspath(__file__)]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def library_hash(path=None):
    """The source hash compiled into a built library, read from the file's bytes (no dlopen), or None."""
    path = path or LIB
    if not os.path.exists(path):
        return None
    data = open(path, "rb").read()
    i = data.find(b"BN_SOURCE_HASH=")
    if i < 0:
        return None
    return data[i + 15:i + 31].decode("ascii", "replace")


def needs_build():
    """True unless the built library carries the hash of the sources beside it (content, not mtimes: a copied tree keeps neither
    its timestamps nor their order)."""
    return library_hash() != source_hash()


# extra hipcc flags per source file (basename), on top of FLAGS.  The weight-gradient kernels gain from the max-ILP machine
# scheduler (fp16 256-tile kernel -13 %, skinny kernel -16 % with analytic normals); the chain kernels lose 1-6 % with it
# (profiles/history/r02_ablation.txt, session 24) - hence their own translation unit.
# The backward chain and the analytic-normal chains gain 1-3 % from the AMDGPU register-pressure trackers in the scheduler (the
# forward loses 2-7 % with them, the weight-gradient kernels 2-20 %: session 25).
# (The weight-fragment prefetch depth of the chain GEMM is a template parameter per kernel instantiation: field_kernels.h
# FwdDepth / BwdDepth.)
# -fno-slp-vectorize (round 3): hipcc's SLP pass packs the epilogues' independent fp32 adds / multiplies into v_pk_*_f32 with
# v_mov shuffles around them - more instructions AND more live registers: the fp16 training forward went from 944 to 132
# bytes of scratch per lane, the inference forward from 700 to 0 (profiles/history/r03_kernel_resources.txt).
_NOSLP = ("-fno-slp-vectorize",)
_TRACKERS = ("-mllvm", "-amdgpu-use-amdgpu-trackers=1") + _NOSLP
FILE_FLAGS = {"field_wgrad.hip": ("-mllvm", "-amdgpu-sched-strategy=max-ilp"),
              "field_fwd.hip": _NOSLP,
              "field_bwd.hip": _TRACKERS, "field_adjoint.hip": _TRACKERS, "field_adjbwd.hip": _TRACKERS}


def build(verbose=False, force=False, defines=(), out=None, extra_flags=(), tag=None, file_flags=None):
    """defines/out/extra_flags/tag: build a VARIANT library (A/B experiments, profiles/): objects go to build/<tag>/, the result
    to `out`; load it with BRDFNERF_HIP_LIB=<out>.  extra_flags: hipcc flags for every file (e.g. -mllvm options)."""
    variant = bool(defines) or out is not None or bool(extra_flags) or bool(file_flags)
    per_file = dict(FILE_FLAGS, **(file_flags or {}))     # file_flags: A/B override of one file's extra flags ({basename: (flags...)})
    if not variant and not force and not needs_build():
        return LIB
    tag = tag or "_".join(d.replace("=", "-") for d in defines) or "default"
    objdir = os.path.join(HERE, "build", tag) if variant else os.path.join(HERE, "build")
    out = out or (os.path.join(objdir, "libbrdfnerf_hip.so") if variant else LIB)
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    sh = source_hash()
    hdr_bytes = b"".join(open(f, "rb").read() for f in sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
                         + [os.path.join(ROOT, "include", "brdfnerf_hip.h")])
    keys = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        stamp = os.path.basename(src) == "error.cpp"            # carries the source hash (so it is recompiled whenever any source changed)
        cmd = [HIPCC] + FLAGS + list(per_file.get(os.path.basename(src), ())) + list(extra_flags) + ["-D" + d for d in defines] + \
            (['-DBN_SOURCE_HASH="%s"' % sh] if stamp else []) + ["-x", "hip", "-c", src, "-o", obj]
        # an object is reused when it was compiled from these bytes with this command line (content key, not mtimes)
        key = hashlib.sha256(open(src, "rb").read() + hdr_bytes + " ".join(cmd).encode()).hexdigest()
        keys.append((obj + ".key", key))
        if os.path.exists(obj) and not force and os.path.exists(obj + ".key") and open(obj + ".key").read() == key:
            continue
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src}\n{log}\n")
        elif verbose:
            keep = [l for l in log.splitlines() if any(k in l for k in ("Function Name", "VGPRs:", "Spill", "ScratchSize",
                                                                       "warning", "LDS Size", "Occupancy"))]
            print("\n".join(keep))
    if failed:
        raise RuntimeError("hipcc failed")
    for path, key in keys:
        open(path, "w").write(key)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")]
    xf = [w for a in sys.argv[1:] if a.startswith("--flags=") for w in a.split("=", 1)[1].split()]
    tags = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--tag=")]
    # --file-flags=field_bwd.hip:-fno-slp-vectorize ...   (replaces FILE_FLAGS of that file; needs --tag)
    ff = {a.split("=", 1)[1].split(":", 1)[0]: tuple(a.split("=", 1)[1].split(":", 1)[1].split()) for a in sys.argv[1:] if a.startswith("--file-flags=")}
    print(build(file_flags=ff or None, verbose="--verbose" in sys.argv, force="--force" in sys.argv, defines=defs, out=outs[0] if outs else None,
                extra_flags=xf, tag=tags[0] if tags else None))
