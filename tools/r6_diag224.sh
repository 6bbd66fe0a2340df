#!/bin/bash
export TRUNK_CASE=r50enc_b8_s224_div
O=gpurun_out
echo "== default"; timeout -k 10 300 python tools/trunk_diag.py > $O/r6_diag224_default.txt 2>&1; head -3 $O/r6_diag224_default.txt | grep -v Warn
echo "== FOLD_BN3_FWD=0"; MSFWSI_ENGINE=fold_bn3_fwd=0,fold_ds_fwd=0 timeout -k 10 300 python tools/trunk_diag.py > $O/r6_diag224_nofwdfold.txt 2>&1; head -3 $O/r6_diag224_nofwdfold.txt | grep -v Warn
echo "== all folds off"; MSFWSI_ENGINE=fold_bn3=0,fold_bn3_fwd=0,fold_ds=0,fold_ds_fwd=0,fold_ds_strided=0 timeout -k 10 300 python tools/trunk_diag.py > $O/r6_diag224_nofold.txt 2>&1; head -3 $O/r6_diag224_nofold.txt | grep -v Warn
echo "== seed 1 default"; TRUNK_SEED=1 TRUNK_BRIEF=1 timeout -k 10 300 python tools/trunk_diag.py 2>&1 | grep -v Warn | head -3
echo "== seed 1 all folds off"; TRUNK_SEED=1 TRUNK_BRIEF=1 MSFWSI_ENGINE=fold_bn3=0,fold_bn3_fwd=0,fold_ds=0,fold_ds_fwd=0,fold_ds_strided=0 timeout -k 10 300 python tools/trunk_diag.py 2>&1 | grep -v Warn | head -3
echo "== 64x64 case default (reference point)"; TRUNK_CASE=r50enc_b16_s64_div TRUNK_BRIEF=1 timeout -k 10 300 python tools/trunk_diag.py 2>&1 | grep -v Warn | head -3
