cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in t256x128P two80 persist1 t512x128; do scripts/power_probe.sh $v 64 64 256 128 1 0; done
for v in t256x128P two80 persist1 t512x128; do scripts/power_probe.sh $v 64 64 128 128 1 0; done
