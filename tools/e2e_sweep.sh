set -e
export E2E_J=16 E2E_REPEAT=5 E2E_KEEP=1
for args in "--contexts 1" "--contexts 2" "--contexts 3"; do
  echo "== $args"
  E2E_ARGS="$args" timeout -k 10 300 python tools/e2e.py 4000 2>&1 | grep -v "^wrote" | tail -2
done
rm -f /dev/shm/e2e.m5* /dev/shm/out.fa
