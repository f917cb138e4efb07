for cfg in "128 2048" "256 2048" "512 2048" "1024 2048" "1024 1024" "1024 512" "1024 256" "1024 128"; do
  set -- $cfg
  python tools/eq_sweep.py --strips $1 --ticks $2 --toggle --steps 8 --chunks 0 2>/dev/null | sed 's/toggle=True fast=False: //'
  python tools/eq_sweep.py --strips $1 --ticks $2 --toggle --steps 8 --chunks 0 --overlap-tail 2>/dev/null | sed 's/toggle=True fast=False: //'
done
