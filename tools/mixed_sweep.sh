export SKIP_PER_EX=1
for ws in 1 2 3; do
  echo "== table, walk streams $ws"
  MFA_MIXED_WALK_STREAMS=$ws ENGINES=table timeout -k 10 200 python tools/walk_check.py timing 2>&1 | grep "^mixed"
done
echo "== jit"; ENGINES=jit timeout -k 10 200 python tools/walk_check.py timing 2>&1 | grep "^mixed"
