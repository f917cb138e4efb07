# planner calibration grid: device time of the eq_three group for forced chunk counts (whole-tick chunks: only some counts are distinct)
for n in 128 256 512 1024; do
  python tools/eq_sweep.py --strips $n --ticks 2048 --toggle --steps 6 --chunks 0,64,128,192,256,344,512,683,1024,2048 2>/dev/null | sed 's/toggle=True fast=False: //'
done
for t in 128 256 512 1024; do
  python tools/eq_sweep.py --strips 1024 --ticks $t --toggle --steps 10 --chunks 0,64,128,256,512 2>/dev/null | sed 's/toggle=True fast=False: //'
done
