cd _old_r02 && python tools/config4.py --resident --frames 128 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r02 code', d['seconds'])"
cd .. && python tools/config4.py --resident --frames 128 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HEAD', d['seconds'])"
