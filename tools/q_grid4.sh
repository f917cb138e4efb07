python tools/eq_sweep.py --strips 1024 --ticks 2048 --toggle --steps 8 --chunks 0,64,128,192,256 2>/dev/null | sed 's/toggle=True fast=False: //'
python tools/eq_sweep.py --strips 1024 --ticks 1024 --toggle --steps 8 --chunks 0,64,128,256 2>/dev/null | sed 's/toggle=True fast=False: //'
MX_EQ_SPEC_SB=321 python tools/eq_sweep.py --strips 1024 --ticks 2048 --toggle --steps 8 --chunks 128 2>/dev/null | sed 's/toggle=True fast=False: /SB=321 /'
MX_EQ_SPEC_SB=32 python tools/eq_sweep.py --strips 1024 --ticks 2048 --toggle --steps 8 --chunks 256 2>/dev/null | sed 's/toggle=True fast=False: /SB=32 /'
