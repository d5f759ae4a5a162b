#!/bin/bash
set +e
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_classifier.py -q -m gpu -x -k "off_the_golden" 2>&1 | tail -12
