python tools/vleg.py 1920 3 main 2>/dev/null | python -c "import sys,json; print('main ', [json.loads(l)['device_us_per_frame'] for l in sys.stdin if l.startswith('{')])"
python tools/vleg.py 1920 2 alpha 2>/dev/null | python -c "import sys,json; print('alpha', [json.loads(l)['device_us_per_frame'] for l in sys.stdin if l.startswith('{')])"
