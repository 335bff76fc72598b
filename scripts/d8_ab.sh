cd /root/repo
for rep in 1 2; do for m in 0 32 44; do SM=1 GN=4 tools/abl_dma8_n2_$m 64 768 768; done; done
