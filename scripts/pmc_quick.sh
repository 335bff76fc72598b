# quick PMC pass over a 5-step pass (usage: bash scripts/pmc_quick.sh "<counters>" <tag>)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/raw
rocprofv3 --pmc $1 -M --output-format csv -d /tmp/pmc_$2 -- python bench.py --ddim-steps 5 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2> gpurun_out/pmc_$2.err
gzip -c /tmp/pmc_$2/*/*counter_collection.csv > gpurun_out/raw/pmc_$2.csv.gz
