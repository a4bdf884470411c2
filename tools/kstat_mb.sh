R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for MB in 8 16 32 64; do
  O=/tmp/kstat_mb$MB; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace -d $O -o r -- python $R/tools/phase_profile.py --mb $MB --reps 2 > $O/log.txt 2>&1
  echo "MB=$MB (max_us column = full-resolution launch; per frame = max_us / MB)"
  python $R/tools/rocpd_stats.py $O/r_results.db | grep -E "front2<2, 3|fed_pair<3>|cand2<3" | awk -v mb=$MB '{printf "%-50s max_us %9.2f  per_frame_us %7.3f\n", $1" "$2" "$3" "$4" "$5, $(NF-1), $(NF-1)/mb}'
done
