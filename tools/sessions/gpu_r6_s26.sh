# Round 6, GPU session 26: the four-stage K = 32 twins of tiles 21 / 17 (ids 27 / 28; built only under PCDM_DEV_KB32) judged IN THE STEP on the
# weight-heavy, latency-bound launches of UNet levels 2 / 3 -- back to back (rounds 2 / 3) they were 10-15 % slower, but there the weights were hot;
# in the step a level-3 convolution waits ~2 us per K-tile on a two-stage ring (tools/probe_cold_launch.py).  Nothing is written: report only.
set -u
OUT=gpurun_out/r6_s26
mkdir -p $OUT
export PCDM_LIB=$PWD/pcdms_amd/lib_alt/kb32/libpcdm.so PCDM_DEV_KB32=1
(timeout 900 python tools/tune_in_step.py --only 704,1280 --out $OUT/tune_704.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_704.txt
(timeout 900 python tools/tune_in_step.py --only 2816,1280 --out $OUT/tune_2816.json 2>&1 | grep -v amdgpu.ids) > $OUT/tune_2816.txt
cut -c1-230 $OUT/tune_704.txt | tail -12; cut -c1-230 $OUT/tune_2816.txt | tail -22
