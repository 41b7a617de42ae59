#!/bin/bash
# round 6, call O: the tiled VAE decode / encode of a WHOLE video (9 tiles x 21 latent frames) against the reference's arithmetic on torch-ROCm
O=gpurun_out/r06
mkdir -p $O
( time timeout 1100 python tests/fullsize_vae_parity.py --out $O/fullsize_vae_parity.json ) > $O/fullsize_vae_parity.log 2>&1
echo "rc=$?" >> $O/fullsize_vae_parity.log; tail -8 $O/fullsize_vae_parity.log | cut -c1-900
( time timeout 600 python tests/fullsize_vae_parity.py --peaky 6 --no-encode --out $O/fullsize_vae_parity_peaky6.json ) > $O/fullsize_vae_parity_peaky6.log 2>&1
echo "rc=$?" >> $O/fullsize_vae_parity_peaky6.log; tail -6 $O/fullsize_vae_parity_peaky6.log | cut -c1-900
