set -x
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warning | tail -12
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
