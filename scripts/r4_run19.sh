cd ${GRAFT_REPO_ROOT:-/root/repo}
ONLY=t256x128P,t512x128S,t512x128FS ROUNDS=4 IT=10 timeout 600 tools/abl_conv_bench256 64 64 128 128 1 256 2>&1 | cut -c1-400
ONLY=t256x128P,t512x128S,t512x128FS ROUNDS=4 IT=10 timeout 600 tools/abl_conv_bench256 64 64 128 128 1 384 2>&1 | cut -c1-400
ONLY=t256x128P,t512x128S,t512x128FS ROUNDS=4 IT=10 timeout 600 tools/abl_conv_bench256 64 32 128 128 1 256 2>&1 | cut -c1-400
