mkdir -p gpurun_out/fuzz_r3
python -m pytest tests -m gpu -x -q > gpurun_out/fuzz_r3/pytest.log 2>&1; echo "pytest rc $?" 
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > gpurun_out/fuzz_r3/$6.log 2>&1; grep -c " ok" gpurun_out/fuzz_r3/$6.log; grep MISMATCH gpurun_out/fuzz_r3/$6.log | head -3; }
t 240 fuzz_structural.py 7000000 7003000 "" structural
FUZZ_REAL=1 t 240 fuzz_structural.py 7100000 7103000 "" real
t 240 fuzz_repeats.py 7200000 7203000 "" repeats
t 120 fuzz_repeats.py 7300000 7301000 rep rep
t 300 fuzz_repeats.py 7400000 7404000 batch batch
t 120 fuzz_repeats.py 7500000 7501000 big big
SKDER_AMD_QUEUES=2 t 200 fuzz_repeats.py 7800000 7803000 batch batch_two_queues
SKDER_AMD_QUEUES=3 t 100 fuzz_repeats.py 7900000 7902000 batch batch_three_queues
t 120 fuzz_repeats.py 7700000 7702000 append append
t 100 fuzz_dropin.py 7600000 7601000 "" dropin
python bench.py --no-realistic 2>/dev/null | tail -1 > gpurun_out/fuzz_r3/bench.json; python -c "
import json; d=json.load(open('gpurun_out/fuzz_r3/bench.json')); print(d['ms_per_step'], d['value'], d.get('phases_ms'))"
