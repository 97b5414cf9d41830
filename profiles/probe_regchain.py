"""Probe: sigma-only trunk with the activations held in registers (profiles/probes/regchain_sigma.hip) against the
product sigma-only kernel (bn_field_sigma) on the bench shape: same weights, same points; prints max |diff| and ms / TFLOP/s of
both.  Not part of the product path.   python profiles/probe_regchain.py [--build-only]"""
import ctypes as C
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "profiles", "probes", "regchain_sigma.hip")
OUT = os.path.join(ROOT, "brdf_nerf_amd", "build", "probe")
DEFS = [a for a in sys.argv[1:] if a.startswith("-D")]
LIB = os.path.join(OUT, "libregchain_probe" + "".join("_" + d[2:] for d in DEFS) + ".so")


def build():
    os.makedirs(OUT, exist_ok=True)
    if os.path.exists(LIB) and os.path.getmtime(LIB) > os.path.getmtime(SRC):
        return
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", "-save-temps=obj",
           "-Rpass-analysis=kernel-resource-usage", "-o", LIB, SRC] + DEFS
    r = subprocess.run(cmd, capture_output=True, text=True)
    print("\n".join(l.split("]")[0].split(":0:")[-1] for l in (r.stdout + r.stderr).splitlines()
                    if ("remark" in l and any(k in l for k in ("VGPRs", "AGPRs", "Scratch", "Spill"))) or "error" in l))
    assert r.returncode == 0, r.stderr[-3000:]


def pack(model, dev):
    """Weight stream in consumption order + pre-scaled biases + permuted sigma weights (layout: see the kernel header)."""
    import torch
    F, NT = 512, 16
    lane = torch.arange(64)
    i, h = lane & 31, lane >> 5
    e = torch.arange(8)
    frags = []
    for l in range(8):
        W = model.fc_net[2 * l].weight.detach().float().cpu()
        scale = (30.0 if l == 0 else 1.0) / (2 * math.pi)
        cols = []      # per k-step: [64 lanes][8] input column (or -1)
        if l == 0 or l == 4:
            for s in range(4):
                p = 16 * s + 8 * h[:, None] + e[None, :]
                cols.append(torch.where(p < 60, p, torch.full_like(p, -1)))
        if l > 0:
            off = 60 if l == 4 else 0
            for s in range(32):
                phi = 32 * (s >> 1) + 16 * (s & 1) + 8 * (e[None, :] >> 2) + 4 * h[:, None] + (e[None, :] & 3)
                cols.append(phi + off)
        cols = torch.stack(cols)                               # [KS][64][8]
        KS = cols.shape[0]
        Wz = torch.cat([W, torch.zeros(F, 1)], 1)              # column -1 -> 0
        rows = (torch.arange(NT)[:, None] * 32 + i[None, :])   # [NT][64]
        fr = Wz[rows[:, None, :, None], cols[None, :, :, :]] * scale     # [NT][KS][64][8]
        fr = fr.view(NT // 2, 2, KS, 64, 8).permute(0, 2, 1, 3, 4)      # [np][ks][t][lane][e]
        frags.append(fr.reshape(-1, 64, 8))
    stream = torch.cat(frags).to(torch.bfloat16).contiguous()
    assert stream.shape[0] == 3712, stream.shape
    bias = torch.stack([model.fc_net[2 * l].bias.detach().float().cpu() * ((30.0 if l == 0 else 1.0) / (2 * math.pi))
                        for l in range(8)]).contiguous()
    sw_full = model.sigma_from_xyz[0].weight.detach().float().cpu().view(-1)
    q = torch.arange(32)
    sw = torch.empty(2, 256)
    for hh in range(2):
        phi = 32 * (q[:, None] >> 1) + 16 * (q[:, None] & 1) + 8 * (e[None, :] >> 2) + 4 * hh + (e[None, :] & 3)
        sw[hh] = sw_full[phi].reshape(-1)
    sb = float(model.sigma_from_xyz[0].bias.detach())
    return stream.to(dev), bias.to(dev), sw.contiguous().to(dev), sb


if __name__ == "__main__":
    build()
    if "--build-only" in sys.argv:
        sys.exit(0)
    import torch
    import bench
    from brdf_nerf_amd import load_model
    from brdf_nerf_amd import functions as Fn
    dev = torch.device("cuda", 0)
    lib = C.CDLL(LIB)
    lib.bn_probe_regchain_sigma.restype = C.c_int
    lib.bn_probe_regchain_sigma.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p]
    args = bench.make_args(4096, 64, 64, "bf16")
    torch.manual_seed(0)
    model = load_model(args).to(dev)
    spec = model.spec(False, False, False)
    packed = model.repack(spec)
    M = 4096 * 128
    xyz = (torch.rand(M, 3, device=dev) * 2 - 1).contiguous()
    stream, bias, sw, sb = pack(model, dev)
    out = torch.zeros(M, device=dev)
    dbg = torch.zeros(256 * 4 * 8, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def probe(n=M, blocks=256):
        rc = lib.bn_probe_regchain_sigma(xyz.data_ptr(), n, stream.data_ptr(), bias.data_ptr(), sw.data_ptr(), sb, out.data_ptr(), blocks, st, dbg.data_ptr())
        assert rc == 0, rc

    ref = Fn.field_sigma(spec, model.named(), packed, xyz=xyz)
    probe(128 * 8, 8)
    torch.cuda.synchronize()
    print("small: max |probe - product| =", float((out[:1024] - ref[:1024]).abs().max()), " max |ref| =", float(ref[:1024].abs().max()))
    probe()
    torch.cuda.synchronize()
    d = (out - ref).abs()
    print("full : max |probe - product| =", float(d.max()), " mean", float(d.mean()), " max |ref| =", float(ref.abs().max()))
    flops = 2.0 * M * (64 * 512 + 6 * 512 * 512 + 576 * 512)

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    t_ref = timed(lambda: Fn.field_sigma(spec, model.named(), packed, xyz=xyz))
    t_pr = timed(probe)
    if any("PROBE_TIMING" in d for d in DEFS):
        probe()
        torch.cuda.synchronize()
        d = dbg.view(256, 4, 8).double().mean((0, 1))
        tot = float(d[:4].sum())
        print("cycles per wave (mean): total %.0f | tile setup+PE %.0f | layer 0 %.0f | layers 1-7 %.0f | drain+sigma head %.0f | "
              "inside fences (wait+barrier) %.0f over %d fences" % (tot, d[0], d[1], d[2], d[3], d[4], int(d[5])))
    print(f"product sigma-only kernel: {t_ref:.3f} ms  {flops / t_ref * 1e-9:.0f} TFLOP/s")
    print(f"register-chain probe     : {t_pr:.3f} ms  {flops / t_pr * 1e-9:.0f} TFLOP/s")
