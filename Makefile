# Convenience targets (everything is also reachable through __graft_entry__.py / pytest / bench.py directly).
PY ?= python

build:            ## hipcc for gfx950 (cross-compiles without a GPU) + the CPU oracle
	$(PY) -c "import __graft_entry__ as g; g.build()"

test:             ## oracle vs reference goldens, C ABI, host logic (no GPU needed)
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu:         ## bit-exact GPU parity (MI355X)
	$(PY) -m pytest tests -x -q -m gpu

smoke:
	$(PY) -c "import __graft_entry__ as g; g.build(); g.smoke()"

bench:            ## one JSON line: env steps/s, roofline, cpu_baseline
	$(PY) bench.py

pin-oracle:       ## build container only: diff the oracle against the imported reference, regenerate nothing
	$(PY) -m tools.oracle.check_oracle_vs_reference --games 36

.PHONY: build test test-gpu smoke bench pin-oracle
