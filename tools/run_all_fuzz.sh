# every randomised cross-check once, on the current tree (about four minutes of GPU time); logs under gpurun_out/
cd $GRAFT_REPO_ROOT
for job in "fuzz_lap 200 50000" "fuzz_support 150 50000" "fuzz_misc 150 50000" "fuzz_float 150 50000" "fuzz_step 200 50000" "fuzz_train 40 50000" "fuzz_pipeline 80 50000"; do
  set -- $job
  timeout -k 10 400 python -u tests/$1.py $2 $3 > gpurun_out/all_$1.log 2>&1 || { echo "FAILED $1"; tail -n 5 gpurun_out/all_$1.log | cut -c1-300; exit 1; }
  echo "$1: $(tail -n 1 gpurun_out/all_$1.log | cut -c1-260)"
done
