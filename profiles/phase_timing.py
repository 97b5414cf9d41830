"""Phase-cycle breakdown of the fused forward chain kernel (diagnostic build, not the product library).

Compiles csrc/field_fwd.hip with -DBN_PHASE_TIMING into brdf_nerf_amd/build/libbn_timing.so (all other objects as built),
runs the sigma-only inference kernel and the training forward on the bench shape, and prints per-wave average shader
cycles per phase (pe, trunk gemm, barriers, epilogue, stash copy, sigma head, feats, heads).
"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from brdf_nerf_amd import build as B

TIMED = ("field_fwd.hip", "field_bwd.hip", "field_wgrad.hip")


def build_timing(defines=(), tag=""):
    B.build()
    objdir = os.path.join(B.HERE, "build")
    objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in B.sources() if not s.endswith(TIMED)]
    procs = []
    for src in TIMED:
        obj = os.path.join(objdir, f"{src}.timing{tag}.o")
        objs.append(obj)
        procs.append(subprocess.Popen([B.HIPCC] + B.FLAGS + list(B.FILE_FLAGS.get(src, ())) + ["-DBN_PHASE_TIMING"] + ["-D" + d for d in defines] +
                                      ["-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj]))
    assert all(p.wait() == 0 for p in procs)
    lib = os.path.join(objdir, f"libbn_timing{tag}.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    defines = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    tag = "".join("_" + d.replace("=", "") for d in defines)
    lib = build_timing(defines, tag)
    print("variant:", defines or "default")
    if "--build-only" in sys.argv:
        sys.exit(0)
    from brdf_nerf_amd import _lib
    _lib.LIB_PATH = lib
    import torch
    import bench
    from brdf_nerf_amd import load_model
    from brdf_nerf_amd import functions as Fn
    L = _lib.lib()
    wg = "BN_PHASE_TIMING_WGRAD" in defines
    for fn in (L.bn_debug_phase_read_fwd, L.bn_debug_phase_read_bwd) + ((L.bn_debug_phase_read_wgrad,) if wg else ()):
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda", 0)
    args = bench.make_args(4096, 64, 64, "bf16")
    torch.manual_seed(0)
    model = load_model(args).to(dev)
    spec = model.spec(False, False, False)
    packed = model.repack(spec)
    b = bench.synthetic_batch(4096, 1, dev)
    z = torch.sort(torch.rand(4096, 128, device=dev) * 2, -1)[0]
    FWD = ["pe", "gemm", "bar_gemm", "epilogue", "bar_epi", "stash", "sigma", "feats_gemm", "feats_epi+stash", "head_gemm",
           "head_epi", "head_reduce", "pp_wait_h0", "pp_wait_h1"]
    # (round 4, barrier-free trunk: 10 = wait until every wave has read this group's columns, 12 = wait for the columns the
    # next half-GEMM reads + the lag of group 1; under barriers: the two workgroup barriers of a layer)
    BWD = ["seed", "head_dG", "bar", "dG_stash", "head_gemm", "bar", "dfeats_epi", "bar", "dfeats_stash", "gemm", "wait_readers",
           "epilogue", "wait_writers", "tail"]
    buf = (ctypes.c_ulonglong * 17)()

    def report(tag, fn, reader, names):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        reader(buf, 1)
        fn()
        torch.cuda.synchronize()
        reader(buf, 1)
        waves = max(1, buf[16])
        tot = sum(buf[i] for i in range(16))
        print(tag, "waves", waves, "cycles/wave", tot // waves)
        for i, n in enumerate(names):
            if n:
                print(f"   {i:2d} {n:16s} {buf[i] / waves:10.0f}  {100.0 * buf[i] / max(1, tot):5.1f} %")

    report("sigma-only (inference)", lambda: Fn.field_sigma(spec, model.named(), packed, rays=b["rays"], z=z),
           L.bn_debug_phase_read_fwd, FWD)
    n = z.numel()
    out = torch.empty(n, spec.out_channels, device=dev)
    stash = torch.empty(Fn.field_stash_bytes(spec, n), dtype=torch.uint8, device=dev)
    report("training forward (stash)", lambda: Fn.field_forward_raw(spec, model.named(), packed, out, stash, rays=b["rays"], z=z),
           L.bn_debug_phase_read_fwd, FWD)
    d_out = torch.randn_like(out)
    grads = {k: torch.zeros_like(v) for k, v in model.named().items()}
    if "BN_PHASE_TIMING_WGRAD" in defines:
        BWD = ["stage: fragment reads + MFMAs + next stage's LDS stores / global loads", "", "", "barrier", "slab stores"] + [""] * 9 + ["prologue"]
    report("backward: wgrad256" if "BN_PHASE_TIMING_WGRAD" in defines else "backward chain", lambda: Fn.field_backward_raw(spec, model.named(), grads, packed, out, d_out, stash, rays=b["rays"], z=z),
           L.bn_debug_phase_read_wgrad if wg else L.bn_debug_phase_read_bwd, BWD)
