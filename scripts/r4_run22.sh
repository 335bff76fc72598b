cd ${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showpower --showclocks 2>&1 | grep -E "sclk|Power" | head -4
for rep in 1 2; do
for v in t256x128P t512x128; do scripts/power_probe.sh $v 64 64 256 128 1 0; done
done
for v in t256x128P t512x128; do scripts/power_probe.sh $v 64 64 128 128 1 0; done
for v in t256x128P t256x256; do IT=20000 scripts/power_probe.sh $v 64 32 512 256 1 0; done
rocm-smi --showmaxpower 2>&1 | grep -i -E "max|power" | head -4
