cd /tmp && export TMPDIR=/tmp
for sl in 16 12 24; do
rm -rf /tmp/kt5 && COSLAM_SYRK_SLICES=$sl timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o b -- python $GRAFT_REPO_ROOT/tools/ba_cfg5_time.py 1 > /tmp/kt5.log 2>&1; DB=$(find /tmp/kt5 -name "*.db" | head -1); echo "slices $sl"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB /tmp/x.md | grep -E "k_syrk_mfma|k_schur_diag_u|k_syrk_reduce|k_cholflow "; tail -3 /tmp/kt5.log | grep SYRK
done
