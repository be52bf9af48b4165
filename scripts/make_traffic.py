"""profiles/traffic.json from a PMC summary (scripts/parse_pmc.py): HBM bytes per launch of each bench stage.

Rule (MI355X_MICROARCH.md, HBM section): bytes = FETCH_SIZE*1024 + WRITE_SIZE*1024, with FETCH_SIZE doubled for kernels whose
reads are wide coalesced streams (16 B per lane: this rocprofv3 tallies their 128-B requests at 64 B).  Kernels that read
with narrow or scattered accesses (the grid gathers; the record scatter reads 12 + 4 B per lane) keep the raw figure.
Usage: python scripts/make_traffic.py gpurun_out/<tag>/pmc_summary.json > profiles/traffic.json"""
import json, sys
summary = json.load(open(sys.argv[1]))
STAGES = {  # stage -> [(kernel-name fragment, reads are wide streams?)]
    "grid_forward": [("k_grid_forward", False)],
    "mlp_forward": [("k_mlp_forward", True)],
    "loss": [("k_loss", True)],
    "mlp_backward": [("k_mlp_transpose_weights", True), ("k_mlp_backward", True), ("k_mlp_finalize_gradients", True)],
    "mlp_train_fused": [("k_mlp_transpose_weights", True), ("k_mlp_train", True), ("k_mlp_finalize_gradients", True)],
    "grid_backward_scatter": [("k_grid_bucket_scatter", False)],
    "grid_backward": [("k_grid_backward_sliced", True), ("k_grid_bucket_owner", True)],
    "adam": [("k_adam_step", True)],
}
fused = any("k_mlp_train" in name for name in summary)  # training_step ran the fused network kernel: the three-kernel stages did not run
out = {}
for stage, kernels in STAGES.items():
    if fused and stage in ("mlp_forward", "loss", "mlp_backward"):
        continue
    total = 0.0
    for frag, wide in kernels:
        for name, cs in summary.items():
            if frag in name and "FETCH_SIZE" in cs:
                total += ((2 if wide else 1) * cs["FETCH_SIZE"] + cs.get("WRITE_SIZE", 0.0)) * 1024
    out[stage] = total
out["_source"] = ("separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; scripts/gpu_pmc.sh, summarised by scripts/parse_pmc.py from " + sys.argv[1] +
                  ") of `bench.py --steps 10 --warmup 3`; bytes per launch, FETCH_SIZE doubled for wide coalesced readers per MI355X_MICROARCH.md; not measured in the bench run that quotes them")
json.dump(out, sys.stdout, indent=1)
