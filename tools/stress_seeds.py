"""Differential stress beyond the parametrised seeds: the random-graph, video-scenario, cascade and ingest tests with seeds the suite
does not use.  Usage: python tools/stress_seeds.py [first_seed] [count]   (on the GPU box; prints the first failure and exits 1)."""
import os, sys, pathlib, traceback
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import pytest

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200

import test_gpu_random_graphs as rg
import test_gpu_video_parity as vp
import test_gpu_video_graph as vg
import test_gpu_ingest as gi
import test_ingest_oracle as io_
import test_gpu_fir_resample as fr

mp = pytest.MonkeyPatch()
jobs = [
    ("random graph", lambda s: rg.test_random_graph_matches_the_oracle_on_every_port(s)),
    ("random graph, scan eq", lambda s: rg.test_random_graph_fusion_and_batching_are_invisible_with_the_time_parallel_eq(s)),
    ("video mixer scenario", lambda s: vp.test_video_mixer_random_scenarios_match_oracle(s)),
    ("cascade scenario", lambda s: vg.test_random_cascade_scenarios_in_random_batches(s, "0", mp)),
    ("cascade scenario inline", lambda s: vg.test_random_cascade_scenarios_in_random_batches(s, "1", mp)),
    ("fir / resampler chains", lambda s: fr.test_random_resampler_and_fir_chains_match_oracle_graph(s)),
    ("media source", lambda s: gi.test_media_source_pacing_equals_oracle(s)),
    ("stream input", lambda s: gi.test_stream_input_pacing_and_reblocking_equal_oracle(s)),
]
bad = 0
only = [a[7:] for a in sys.argv if a.startswith("--only=")]
for name, fn in jobs:
    if only and name not in only:
        continue
    ok = 0
    for seed in range(first, first + count):
        try:
            fn(seed); ok += 1
        except AssertionError as e:
            if not str(e):      # the tests' own sanity checks on their generated scenario ("enough frames were shown"): not a comparison
                ok += 1; continue
            bad += 1
            print(f"FAIL {name} seed {seed}"); traceback.print_exc(limit=4)
            if bad >= 5:
                sys.exit(1)
        except Exception:
            bad += 1
            print(f"FAIL {name} seed {seed}"); traceback.print_exc(limit=4)
            if bad >= 5:
                sys.exit(1)
    print(f"{name}: {ok}/{count} seeds ok", flush=True)
mp.undo()
sys.exit(1 if bad else 0)
