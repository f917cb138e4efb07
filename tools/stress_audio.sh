#!/bin/bash
# The audio-side differential stress tools (EqThree shapes / silences / contract mode, Envelope, schedules, edits, Mixer, modules, threads, held-back Mixer banks) at seeds the
# suite does not use.   gpurun -- 'bash tools/stress_audio.sh [first seed]'
cd $GRAFT_REPO_ROOT
F=${1:-50000}
for t in "stress_eq_shapes.py $F 120" "stress_eq_silences.py $F 60" "stress_eq_silences.py $F 40 --contract" "stress_env.py $F 60" "stress_schedule.py $F 40" "stress_seeds.py $F 30" "stress_edits.py $F 30" "stress_mixer.py $F 30" "stress_modules.py $F 30" "stress_osc.py $F 400" "stress_threads.py 10" "stress_overlap.py $F 80"; do
  echo "== $t"; timeout 900 python tools/$t 2>&1 | tail -3
done
