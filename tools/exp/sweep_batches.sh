for n in 256 512 768 1024 1280 1526; do
  python bench.py --batches $n --steps 100 --warmup 5 --no-cpu-baseline --no-variants --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print($n, d['ms_per_step'], d['roofline']['kernel_ms'])"
done
