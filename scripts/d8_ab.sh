cd /root/repo
for rep in 1 2; do
SM=1 GN=4 tools/abl_dma8_n2_0 64 768 768
SM=1 GN=4 COLD=1 tools/abl_dma8_n2_0 64 768 768
SM=1 GN=4 COLD=4 tools/abl_dma8_n2_0 64 768 768
SM=1 GN=4 COLD=24 tools/abl_dma8_n2_0 64 768 768
SM=1 GN=1 COLD=24 tools/abl_dma8_n2_0 64 768 768
SM=1 GN=4 COLD=48 tools/abl_dma8_n2_0 64 768 768
done
