python3 -m pytest tests -m gpu -q 2>&1 | tail -4
tools/ab.sh "pre tree" 2
