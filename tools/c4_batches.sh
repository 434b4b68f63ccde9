P='import json,sys; d=json.loads(sys.stdin.read()); print("%.4f s" % d["seconds"])'
for i in 1 2; do for b in 0 64 32 16; do echo -n "batch $b: "; python tools/config4.py --frames 128 --resident --reuse-handles --arith separable --batch $b 2>/dev/null | python -c "$P"; done; done
for e in 8 16 32 64; do echo -n "ecc-batch $e: "; python tools/config4.py --frames 128 --resident --reuse-handles --arith separable --ecc-batch $e 2>/dev/null | python -c "$P"; done
