export TMPDIR=/tmp
for d in 0 1 0 1; do echo DIRECT $d; YUME_GEMM_EPI_DIRECT=$d timeout 200 tools/gemm_check --timing 2>&1 | grep "big/256" | grep "epi=0\|epi=1\|epi=4" | cut -c1-52,95-220; done
for d in 0 1; do echo VAE DIRECT $d; YUME_GEMM_EPI_DIRECT=$d python tools/vae_probe.py 2>&1 | grep "decode\|encode" | tail -2; done
