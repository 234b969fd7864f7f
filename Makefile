# Convenience targets (the driver uses __graft_entry__.build() / pytest / bench.py directly).
PY ?= python

build:            ## libgorp_hip.so for gfx950 (hipcc cross-compiles without a GPU) + the oracle
	$(PY) -c "import __graft_entry__ as g; g.build()"

test: build       ## CPU tests: oracle vs golden vectors, compiler vs oracle, ABI, DSL, sanitizers, gloo
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu: build   ## on an MI355X: kernels vs oracle, bit-exact
	$(PY) -m pytest tests -x -q -m gpu

bench: build      ## the contract line: 10 M x 200 B lines, README 3-extraction definition
	$(PY) bench.py

.PHONY: build test test-gpu bench
