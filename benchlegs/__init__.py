"""The legs of bench.py, one module each: `headline` (BASELINE configs[1], and configs[4] when launched on N ranks), `variants` (the same job under
other flags / material), `realtime` (one tick per submission; the north-star's 10 240 strips + 8 layers), `video` (configs[3]), `fir` (configs[2]),
`scaling` (what a rank of an N-GPU job costs, measured on one GPU), `cpu` (the CPU oracle timed as the baseline), `line` (the compact stdout line).
bench.py parses the flags, runs the legs and prints; everything a leg measures goes to bench_full.json, one number per leg onto the line."""
