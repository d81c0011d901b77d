# ms/step and trailing-update TFLOP/s of bench.py at several N (extra env vars pass through)
for n in ${SIZES:-2048 4096 8192 16384}; do
  st=200; [ $n -ge 8192 ] && st=40; [ $n -ge 16384 ] && st=10; [ $n -ge 32768 ] && st=4
  echo -n "n=$n: "; python bench.py --n $n --no-cpu-baseline --steps $st --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), 'ms', round(d['roofline']['achieved'],2), 'TF')"
done
