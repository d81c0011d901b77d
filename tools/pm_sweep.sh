for pm in 32 30 28 26 24 20; do
  echo "pair_min=$pm"; GPX_PAIR_MIN=$pm python bench.py --no-cpu-baseline --steps 200 --warmup 20 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"
done
for n in 6144 8192; do for pm in 32 40 48; do for qm in 48 64; do
  echo "n=$n pair_min=$pm quad_min=$qm"; GPX_PAIR_MIN=$pm GPX_QUAD_MIN=$qm python bench.py --n $n --no-cpu-baseline --steps 40 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"
done; done; done
