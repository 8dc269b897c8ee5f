export TMPDIR=/tmp; O=gpurun_out/r04_c5b; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "density or cluster" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks5 -- python3 bench.py --config 5 --ncell 63 --steps 200 --warmup 50 --no-cpu-baseline --dropin-steps 0 > $O/c5.json 2> $O/c5.err
python3 tools/kstats.py $(find $O/ks5 -name "*kernel_stats.csv") 8
