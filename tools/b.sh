#!/bin/bash
# rebuild everything in-tree (library, CLI, oracle, synthgen) from any working directory; prints only errors
cd "$(dirname "$0")/.." && python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -iE "error|undefined" -A6 | head -40; exit ${PIPESTATUS[0]}

