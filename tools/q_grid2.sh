for n in 128 256; do python tools/eq_sweep.py --strips $n --ticks 2048 --toggle --steps 6 --chunks 0 2>/dev/null | sed 's/toggle=True fast=False: //'; done
for t in 64 128 256 512 1024 2048; do python tools/eq_sweep.py --strips 1024 --ticks $t --toggle --steps 10 --chunks 0 2>/dev/null | sed 's/toggle=True fast=False: //'; done
