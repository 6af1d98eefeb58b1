#!/bin/bash
# round 3, GPU call E: full GPU suite after the function-gap commit
O=gpurun_out/r03e; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
