set -u
export TMPDIR=/tmp
for v in 1 0 1 0; do
  echo "theta_from_host=$v"; SGP_SMALL_THETA_HOST=$v timeout 600 python3 tools/bench_configs.py 2>/dev/null | head -5 | cut -c1-200
done
timeout 900 python3 -m pytest tests/test_small.py tests/test_composite.py tests/test_nuts_device_logic.py tests/test_posterior_pin.py -x -q -m gpu 2>&1 | tail -3
