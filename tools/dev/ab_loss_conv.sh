for i in 1 2; do
for f in fp32 split; do
echo "== $f"
S2L_BENCH_LOSS_CONV=$f python tools/bench_train.py 64 bf16 --sync=8 --trainbn 2>&1 | tail -1 | grep -o "ms_per_step.: [0-9.]*"
S2L_BENCH_LOSS_CONV=$f python tools/bench_train.py 8 bf16 --full --trainbn 2>&1 | tail -1 | grep -o "ms_per_step.: [0-9.]*"
S2L_BENCH_LOSS_CONV=$f python tools/bench_train.py 8 bf16 --full --early 2>&1 | tail -1 | grep -o "ms_per_step.: [0-9.]*"
S2L_BENCH_LOSS_CONV=$f python tools/bench_train.py 8 bf16 --full 2>&1 | tail -1 | grep -o "ms_per_step.: [0-9.]*"
done; done
