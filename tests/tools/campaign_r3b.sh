mkdir -p gpurun_out/fuzz_r3b
python -m pytest tests -m gpu -x -q > gpurun_out/fuzz_r3b/pytest.log 2>&1; echo "pytest rc $?" 
t() { timeout $1 python tests/tools/$2 $3 $4 $5 > gpurun_out/fuzz_r3b/$6.log 2>&1; grep -c " ok" gpurun_out/fuzz_r3b/$6.log; grep MISMATCH gpurun_out/fuzz_r3b/$6.log | head -3; }
t 240 fuzz_structural.py 9000000 9003000 "" structural
FUZZ_REAL=1 t 240 fuzz_structural.py 9100000 9103000 "" real
t 240 fuzz_repeats.py 9200000 9203000 "" repeats
t 120 fuzz_repeats.py 9300000 9301000 rep rep
t 300 fuzz_repeats.py 9400000 9404000 batch batch
t 120 fuzz_repeats.py 9500000 9501000 big big
SKDER_AMD_QUEUES=2 t 200 fuzz_repeats.py 9800000 9803000 batch batch_two_queues
SKDER_AMD_QUEUES=3 t 100 fuzz_repeats.py 9900000 9902000 batch batch_three_queues
t 120 fuzz_repeats.py 9700000 9702000 append append
t 100 fuzz_dropin.py 9600000 9601000 "" dropin
python bench.py --no-realistic 2>/dev/null | tail -1 > gpurun_out/fuzz_r3b/bench.json; python -c "
import json; d=json.load(open('gpurun_out/fuzz_r3b/bench.json')); print(d['ms_per_step'], d['value'], d.get('phases_ms'))"
