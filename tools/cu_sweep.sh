for r in 0 8 16 24 32 48 64; do
  echo "reserved=$r"; GPX_RESERVED_CUS=$r python bench.py --no-cpu-baseline --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"
done
for r in 8 16 32; do
  echo "Q masked too reserved=$r"; GPX_RESERVED_Q=1 GPX_RESERVED_CUS=$r python bench.py --no-cpu-baseline --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"
done
for r in 0 16 32; do
  echo "n=8192 reserved=$r"; GPX_RESERVED_CUS=$r python bench.py --n 8192 --no-cpu-baseline --steps 40 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"
done
for r in 0 16 32; do
  echo "n=2048 reserved=$r"; GPX_RESERVED_CUS=$r python bench.py --n 2048 --no-cpu-baseline --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"
done
