cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/s3; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 ./build/issue_rates > $OUT/ubench.log 2>&1; echo rc=$? >> $OUT/ubench.log
for v in "base:" "lookback:QOIMI_ENC_LOOKBACK=1" "lookback_noticket:QOIMI_ENC_LOOKBACK=1 QOIMI_ENC_TICKET=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu --encode-only > $OUT/enc_$name.log 2>&1; echo rc=$? >> $OUT/enc_$name.log
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu) > $OUT/prof.log 2>&1
find $OUT/prof -name "*stats*" | head
