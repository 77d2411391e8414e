"""Times the UNMODIFIED reference module on the bench workload in the authoring container - TEST / MEASUREMENT INFRASTRUCTURE ONLY.

    python -m oracle.time_reference_cpu [repeats]      -> profiles/r03/reference_cpu_timing.json

NextDiT_2B_patch2(qk_norm=True, cap_feat_dim=2048).forward_with_cfg, fp32, CPU, all host cores, one NFE of BASELINE configs[1]
(z [2,4,128,128], T = 128, cfg 4, proportional attention), synthetic weights (oracle.synth).  One warm-up, then `repeats` timed calls;
min / median and the 1-minute load average around each call are recorded (a single sample of this number has
read 38 s, 66 s and 101 s on this shared container).  bench.py quotes the file beside its own measured `cpu_baseline`.
"""
import json
import os
import statistics
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import ref_harness as R  # noqa: E402
from oracle import synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    torch.set_grad_enabled(False)
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    cfg = synth.NEXT_2B
    sd = synth.synth_state_dict(cfg, seed=61, streams=True)
    R.load_reference("lumina_next_t2i")
    import importlib
    model = importlib.import_module("models.model").NextDiT(**cfg.ctor_kwargs()).eval()
    model.load_state_dict(sd, strict=True, assign=True)
    z, t, cap, mask = synth.synth_inputs(cfg, latent_hw=(128, 128), text_len=128, uncond_len=8, seed=62, t_value=0.5)
    kw = dict(base_seqlen=4096, proportional_attn=True)
    times, loads = [], []
    for i in range(reps + 1):
        l0 = os.getloadavg()[0]
        t0 = time.time()
        out = model.forward_with_cfg(z, t, cap, mask, 4.0, **kw)
        dt = time.time() - t0
        assert torch.isfinite(out).all()
        print(f"call {i}{' (warm-up)' if i == 0 else ''}: {dt:.1f} s, 1-min load before {l0:.2f}", flush=True)
        if i:
            times.append(dt)
            loads.append(l0)
    best, med = min(times), statistics.median(times)
    rec = {
        "what": "UNMODIFIED /root/reference lumina_next_t2i NextDiT_2B_patch2(qk_norm=True, cap_feat_dim=2048).forward_with_cfg, fp32, CPU, one NFE at "
                "the bench workload (z [2,4,128,128], T=128, cfg 4, proportional attention), synthetic weights (oracle.synth seed 61 streams)",
        "seconds_per_nfe": best, "seconds_per_nfe_median": med, "samples_s": [round(x, 1) for x in times], "load_1min_before_each": loads,
        "latent_tokens_per_s": 4096 / best, "denoising_steps_per_s": 1.0 / best, "cores": cores,
        "host": f"authoring container ({cores} x86 cores shared with other jobs, torch {torch.__version__} CPU)",
        "source": "oracle/time_reference_cpu.py (min of the timed calls; median and per-call host load beside it)",
    }
    out_dir = os.path.join(REPO, "profiles", "r03")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "reference_cpu_timing.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
