"""Probe: shader cycles per MFMA of the chain kernels' tile GEMM (field_kernels.h gemm_range) alone in a kernel
(profiles/probes/gemm_rate.hip): 1 or 2 waves per SIMD, weight-prefetch depth 2 / 4 / 6 / 8, with and without the weight
stream (L2 -> registers) and the LDS fragment reads (a two-set B-fragment variant was measured in round 4 and dropped:
profiles/r04_probe_gemm_rate.txt keeps its rows).
32 cycles per MFMA per SIMD is the matrix pipe's rate.   python profiles/probe_gemm_rate.py [--build-only | --no-build]"""
import ctypes as C
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "profiles", "probes", "gemm_rate.hip")
OUT = os.path.join(ROOT, "brdf_nerf_amd", "build", "probe")
VARIANTS = [(), ("BN_PROBE_NO_A",), ("BN_PROBE_NO_B",), ("BN_PROBE_NO_A", "BN_PROBE_NO_B"),
            ("BN_PROBE_HALVES",), ("BN_PROBE_STREAM",)]
TABLE1 = VARIANTS[:4]


def lib_of(defs):
    return os.path.join(OUT, "libgemm_rate" + "".join("_" + d for d in defs) + ".so")


def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for defs in VARIANTS:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", "-fno-slp-vectorize",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "brdf_nerf_amd", "csrc"), "-Wno-pass-failed",
               "-o", lib_of(defs), SRC] + ["-D" + d for d in defs]
        procs.append(subprocess.Popen(cmd))
    assert all(p.wait() == 0 for p in procs)


if __name__ == "__main__":
    if "--no-build" not in sys.argv:
        build()
    if "--build-only" in sys.argv:
        sys.exit(0)
    import torch
    dev = torch.device("cuda", 0)
    layers, reps, blocks = 8, 6, 256
    packed = (torch.randn(layers * 512 * 512, device=dev) * 0.05).to(torch.bfloat16)
    cyc = torch.zeros(blocks * 8, dtype=torch.int64, device=dev)
    sink = torch.zeros(4, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    print("cycles per MFMA of ONE wave (median over 256 workgroups x waves; 2 waves per SIMD: the pipe serves both, so 64 = pipe-bound;"
          " 1 wave per SIMD: 32 = pipe-bound); 8 layers x 256 MFMAs per wave x 6 passes, weights 4 MB bf16 (L2), LDS tile 128 x 512")
    print(f"{'variant':44s} " + " ".join(f"{'w/SIMD ' + str(w) + ' depth ' + str(d):>16s}" for w in (1, 2) for d in (2, 4, 6, 8)))
    for defs in TABLE1:
        lib = C.CDLL(lib_of(defs))
        lib.bn_probe_gemm_rate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_int]
        row = []
        for wps in (1, 2):
            for depth in (2, 4, 6, 8):
                for _ in range(3):
                    rc = lib.bn_probe_gemm_rate(packed.data_ptr(), layers, reps, wps, depth, cyc.data_ptr(), sink.data_ptr(), blocks, st, None, 0, 0)
                    assert rc == 0, rc
                torch.cuda.synchronize()
                v = cyc.view(blocks, 8)[:, :4 * wps].reshape(-1).tolist()
                row.append(statistics.median(v) / (layers * reps * 256))
        name = "chain GEMM (product)"
        name += "".join({"BN_PROBE_NO_A": ", no weight stream", "BN_PROBE_NO_B": ", no LDS reads"}.get(d, "") for d in defs)
        print(f"{name:44s} " + " ".join(f"{x:16.1f}" for x in row))

    # gemm_stream (continuous weight stream) against two gemm_range calls per layer, with a layer's stash stores behind every layer
    stream = torch.empty(6 << 30, dtype=torch.uint8, device=dev)
    print("\ntwo half-GEMMs per layer (the trunk's shape), 8 layers: cycles per MFMA of one wave; st = stash stores per wave and layer")
    print(f"{'variant':44s} " + " ".join(f"{'w' + str(w) + ' d' + str(d) + ' st' + str(sn):>11s}" for w in (1, 2) for d in (2, 4, 8) for sn in (0, 24)))
    for defs, name in ((("BN_PROBE_HALVES",), "gemm_range x 2 (rounds 1-3)"), (("BN_PROBE_STREAM",), "gemm_stream x 2 (ring carried across layers)")):
        lib = C.CDLL(lib_of(defs))
        lib.bn_probe_gemm_rate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_size_t, C.c_int]
        row = []
        for wps in (1, 2):
            for depth in (2, 4, 8):
                for sn in (0, 24):
                    for _ in range(3):
                        rc = lib.bn_probe_gemm_rate(packed.data_ptr(), layers, reps, wps, depth, cyc.data_ptr(), sink.data_ptr(), blocks, st, stream.data_ptr(), stream.numel(), sn)
                        assert rc == 0, rc
                    torch.cuda.synchronize()
                    v = cyc.view(blocks, 8)[:, :4 * wps].reshape(-1).tolist()
                    row.append(statistics.median(v) / (layers * reps * 256))
        print(f"{name:44s} " + " ".join(f"{x:11.1f}" for x in row))
    # the weight footprint against the 4 MB of an XCD's L2, and a layer's stash stores beside the GEMM (product GEMM, depth 6)
    lib = C.CDLL(lib_of(()))
    lib.bn_probe_gemm_rate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_int]
    print("\nproduct GEMM, depth 6: cycles per MFMA of one wave by weight footprint (layers x 512 KB, swept cyclically by every workgroup) and"
          " stash stores per wave and layer (x 1 KB, non-temporal, streaming; the training forward writes 24)")
    print(f"{'layers (MB of weights)':>24s} " + " ".join(f"{'w/SIMD ' + str(w) + ' st ' + str(sn):>15s}" for w in (1, 2) for sn in (0, 24)))
    for nl in (4, 7, 8, 9, 10, 12, 16):
        pk = (torch.randn(nl * 512 * 512, device=dev) * 0.05).to(torch.bfloat16)
        row = []
        for wps in (1, 2):
            for sn in (0, 24):
                rp = max(2, 48 // nl)
                for _ in range(3):
                    rc = lib.bn_probe_gemm_rate(pk.data_ptr(), nl, rp, wps, 6, cyc.data_ptr(), sink.data_ptr(), blocks, st, stream.data_ptr(), stream.numel(), sn)
                    assert rc == 0, rc
                torch.cuda.synchronize()
                v = cyc.view(blocks, 8)[:, :4 * wps].reshape(-1).tolist()
                row.append(statistics.median(v) / (nl * rp * 256))
        print(f"{nl:>14d} ({nl * 0.5:4.1f} MB) " + " ".join(f"{x:15.1f}" for x in row))
