# tail time by bounce cap with the scene read through scalar loads instead of LDS (experiment)
set -e
RT355_BOUNCE_PASSES=0 RT355_NO_LDS=1 timeout -k 10 120 python tools/exp_tail.py > gpurun_out/exp_tail_nolds.txt 2>&1
