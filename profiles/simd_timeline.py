"""Per-SIMD timeline of the training forward's trunk (diagnostic build, -DBN_TIMELINE).

Every wave of the first 8 workgroups stamps s_memtime at the phase boundaries of the trunk (field_kernels.h BN_TL); the script
prints, for the two waves of each SIMD (wave w and w + 4: HW_ID confirms the SIMD), when each was in a GEMM half and when in an
epilogue pass, per layer, in shader cycles from the workgroup's first stamp - and the share of the trunk during which exactly
one / both / neither of the two waves was inside a GEMM.

    python profiles/simd_timeline.py [-DFLAG ...] [--sigma] [--blocks 2] [--build-only | --no-build]
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from brdf_nerf_amd import build as B  # noqa: E402

EV = {0: "layer", 1: "g0+", 2: "g0-", 3: "g1+", 4: "g1-", 5: "S+", 6: "S-", 7: "C+", 8: "C-", 9: "end", 10: "Sc-"}
NEV, NBLK = 112, 8


def build_tl(defines, tag):
    B.build()
    objdir = os.path.join(B.HERE, "build")
    objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in B.sources() if not s.endswith("field_fwd.hip")]
    obj = os.path.join(objdir, f"field_fwd.hip.timeline{tag}.o")
    subprocess.check_call([B.HIPCC] + B.FLAGS + list(B.FILE_FLAGS.get("field_fwd.hip", ())) + ["-DBN_TIMELINE"] + ["-D" + d for d in defines] +
                          ["-x", "hip", "-c", os.path.join(B.CSRC, "field_fwd.hip"), "-o", obj])
    lib = os.path.join(objdir, f"libbn_timeline{tag}.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + [obj])
    return lib


def intervals(events, a, b):
    """[(t0, t1)] between event codes a (open) and b (close)"""
    out, t0 = [], None
    for code, t in events:
        if code == a:
            t0 = t
        elif code == b and t0 is not None:
            out.append((t0, t))
            t0 = None
    return out


def covered(ivs, lo, hi, step=64):
    n = (hi - lo) // step + 1
    m = [0] * n
    for t0, t1 in ivs:
        for i in range(max(0, (t0 - lo) // step), min(n, (t1 - lo) // step + 1)):
            m[i] = 1
    return m


if __name__ == "__main__":
    defines = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    tag = "".join("_" + d.replace("=", "") for d in defines)
    lib = os.path.join(B.HERE, "build", f"libbn_timeline{tag}.so") if "--no-build" in sys.argv else build_tl(defines, tag)
    if "--build-only" in sys.argv:
        sys.exit(0)
    nblk = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else 2
    from brdf_nerf_amd import _lib
    _lib.LIB_PATH = lib
    import torch
    import bench
    from brdf_nerf_amd import load_model
    from brdf_nerf_amd import functions as Fn
    L = _lib.lib()
    L.bn_debug_timeline_read_fwd.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    args = bench.make_args(4096, 64, 64, "bf16")
    torch.manual_seed(0)
    model = load_model(args).to(dev)
    spec = model.spec(False, False, False)
    packed = model.repack(spec)
    b = bench.synthetic_batch(4096, 1, dev)
    z = torch.sort(torch.rand(4096, 64, device=dev) * 2, -1)[0]
    n = z.numel()
    out = torch.empty(n, spec.out_channels, device=dev)
    stash = torch.empty(Fn.field_stash_bytes(spec, n), dtype=torch.uint8, device=dev)
    sigma = "--sigma" in sys.argv
    run = (lambda: Fn.field_sigma(spec, model.named(), packed, rays=b["rays"], z=z)) if sigma else \
        (lambda: Fn.field_forward_raw(spec, model.named(), packed, out, stash, rays=b["rays"], z=z))
    for _ in range(20):      # the chip under load (the first tiles of every launch are the stamped ones)
        run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (NBLK * 8 * NEV))()
    assert L.bn_debug_timeline_read_fwd(buf) == 0
    print("variant:", defines or "default", "| kernel:", "sigma-only inference forward" if sigma else "training forward (stash)",
          "| 262,144 points, bf16, F = 512; cycles = s_memtime ticks from the workgroup's first stamp")
    print("events: g0 / g1 = GEMM over input-column half 0 / 1, S = pass S (sin, pack, LDS tile + Y stash; 'Sc' = its arithmetic done,",
          "group 0 then waits for the readers), C = pass C (cos, 8-bit D stash)")
    tot = {"one": 0, "both": 0, "none": 0}
    for blk in range(nblk):
        waves = []
        for w in range(8):
            raw = [buf[(blk * 8 + w) * NEV + i] for i in range(NEV)]
            hw = raw[0]
            ev = [(int(x >> 56), int(x & ((1 << 56) - 1))) for x in raw[1:] if x]
            waves.append((hw, ev))
        t0 = min(ev[0][1] for hw, ev in waves if ev)
        print(f"\nworkgroup {blk}")
        for simd_pair in range(4):
            wa, wb = simd_pair, simd_pair + 4
            print(f"  waves {wa} (group 0) and {wb} (group 1): HW_ID simd {(waves[wa][0] >> 4) & 3} / {(waves[wb][0] >> 4) & 3}, cu {(waves[wa][0] >> 8) & 15} / {(waves[wb][0] >> 8) & 15}")
            for w in (wa, wb):
                hw, ev = waves[w]
                line, layer = [], -1
                for code, t in ev:
                    if code == 0:
                        layer += 1
                        line.append(f"\n      L{layer}:")
                    else:
                        line.append(f"{EV.get(code, code)}{t - t0}")
                print(f"    wave {w}:" + " ".join(line))
            if blk == 0 or True:
                ga = intervals(waves[wa][1], 1, 2) + intervals(waves[wa][1], 3, 4)
                gb = intervals(waves[wb][1], 1, 2) + intervals(waves[wb][1], 3, 4)
                if ga and gb:
                    lo = min(ga[0][0], gb[0][0])
                    hi = max(e[1] for e in waves[wa][1] + waves[wb][1] if e[0] == 9)
                    ma, mb = covered(ga, lo, hi), covered(gb, lo, hi)
                    one = sum(1 for x, y in zip(ma, mb) if x + y == 1)
                    both = sum(1 for x, y in zip(ma, mb) if x + y == 2)
                    none = len(ma) - one - both
                    tot["one"] += one; tot["both"] += both; tot["none"] += none
                    print(f"    layers 1-7 ({hi - lo} cycles): one wave in a GEMM {100 * one / len(ma):.1f} %, both {100 * both / len(ma):.1f} %, neither {100 * none / len(ma):.1f} %")
    s = sum(tot.values())
    if s:
        print(f"\nall SIMD pairs of {nblk} workgroups: one wave in a GEMM {100 * tot['one'] / s:.1f} %, both {100 * tot['both'] / s:.1f} %, neither {100 * tot['none'] / s:.1f} %")
