"""How the reduced-precision PSNR gate of BASELINE config 3 (RPV + analytic normals) depends on the STARTING STATE and on how
the continuation is set up (tests/test_gpu_parity.py::test_reduced_precision_heldout_psnr_tracks_fp32_rpv_analytic_normals).

For several independently trained fp32 models (same Lambertian pretraining, BRDF stage with different sampling draws), 150 more
steps in fp32 / bf16 / fp16 with identical batches and draws, in four set-ups: learning rate 1e-4 -> 0 or 2e-5 -> 0, Adam state
fresh or carried over from the training run.  Deterministic mode: every number is reproducible.
    python profiles/psnr_state_study.py [n_states]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import brdf_nerf_amd  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle.config import FieldConfig  # noqa: E402


def main():
    n_states = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    brdf_nerf_amd.set_deterministic(True)
    cfg = FieldConfig(n_samples=64, guided_samples=64, **T.RPV_NAN)
    train, held = T._learnable_table(1024 * 64, 3), T._learnable_table(8192, 11)
    _, first, warm = T._psnr_run(cfg, "fp32", 400, 0, train, held, draw_seed=1)
    print(f"first-step PSNR {first:.2f} dB", flush=True)
    for st in range(n_states):
        adam = {}
        p0, _, trained = T._psnr_run(cfg, "fp32", 0, 600, train, held, draw_seed=3 + 10 * st, init_state=warm, keep_adam=adam)
        for lr0 in (1e-4, 2e-5):
            for carry in (False, True):
                r = {dt: T._psnr_run(cfg, dt, 0, 150, train, held, draw_seed=7, init_state=trained, lr0=lr0, adam=adam if carry else None)[0]
                     for dt in ("fp32", "bf16", "fp16")}
                print(f"state {st} ({p0:.4f} dB)  lr0 {lr0:g}  adam {'carried' if carry else 'fresh  '}:  fp32 {r['fp32']:.4f}  "
                      f"bf16 {r['bf16'] - r['fp32']:+.4f}  fp16 {r['fp16'] - r['fp32']:+.4f}", flush=True)


if __name__ == "__main__":
    main()
