python -m pytest tests/test_gpu_api.py tests/test_gpu_examples.py tests/test_gpu_fullsize.py -q -x -k "jacobian or cahn or system or newton or System or solve" 2>&1 | tail -3
C4_STEPS=12 python tools/c4_step.py 512 2>&1 | grep "STEPS"
C4_STEPS=12 python tools/c4_step.py 512 2>&1 | grep "STEPS"
C4_STEPS=3 bash tools/c4_timeline.sh 2>&1 | grep " us " | tail -14 | cut -c1-150
