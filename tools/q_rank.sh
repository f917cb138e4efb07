# the rank shape of an 8-GPU job (128 strips x 2048 ticks) and the short submission (1024 strips x 64 ticks): plans x overlap
python tools/eq_sweep.py --strips 128 --ticks 2048 --toggle --steps 8 --chunks 0,512,768,1024,1536,2048 2>/dev/null
python tools/eq_sweep.py --strips 128 --ticks 2048 --toggle --steps 8 --chunks 0,768,1024,1536 --overlap-tail 2>/dev/null
python tools/eq_sweep.py --strips 1024 --ticks 64 --toggle --steps 40 --chunks 0 2>/dev/null
python tools/eq_sweep.py --strips 1024 --ticks 64 --toggle --steps 40 --chunks 0 --overlap-tail 2>/dev/null
python tools/eq_sweep.py --strips 256 --ticks 2048 --toggle --steps 6 --chunks 0 2>/dev/null
python tools/eq_sweep.py --strips 256 --ticks 2048 --toggle --steps 6 --chunks 0 --overlap-tail 2>/dev/null
python tools/eq_sweep.py --strips 512 --ticks 2048 --toggle --steps 6 --chunks 0 2>/dev/null
python tools/eq_sweep.py --strips 512 --ticks 2048 --toggle --steps 6 --chunks 0 --overlap-tail 2>/dev/null
