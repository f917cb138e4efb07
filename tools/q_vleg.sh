python tools/vleg.py 1920 3 | python -c "import sys,json; print([json.loads(l)['device_us_per_frame'] for l in sys.stdin if l.startswith('{')])"
