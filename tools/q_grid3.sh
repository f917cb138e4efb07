for n in 128 256 512; do
  python tools/eq_sweep.py --strips $n --ticks 2048 --toggle --steps 6 --chunks 0,256,512,683,1024,2048 2>/dev/null | sed 's/toggle=True fast=False: //'
done
python tools/eq_sweep.py --strips 1024 --ticks 2048 --toggle --steps 6 --chunks 0,192,256,344,512 2>/dev/null | sed 's/toggle=True fast=False: //'
python tools/eq_sweep.py --strips 1024 --ticks 256 --toggle --steps 10 --chunks 0,64,128,256 2>/dev/null | sed 's/toggle=True fast=False: //'
python tools/eq_sweep.py --strips 1024 --ticks 512 --toggle --steps 10 --chunks 0,64,128,256,512 2>/dev/null | sed 's/toggle=True fast=False: //'
