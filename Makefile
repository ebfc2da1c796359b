# Convenience targets; the source of truth is __graft_entry__.build() (hipcc, --offload-arch=gfx950).
PY ?= python

build:
	$(PY) -c "import __graft_entry__ as g; g.build()"

test: build
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu: build            # needs an MI355X
	$(PY) -m pytest tests -q -m gpu

smoke: build               # needs an MI355X
	$(PY) -c "import __graft_entry__ as g; g.smoke()"

bench: build               # needs an MI355X; prints one JSON line
	$(PY) bench.py

goldens:                   # regenerate tests/golden/*.npz from the reference (needs /root/reference)
	$(PY) tests/golden/make_goldens.py

.PHONY: build test test-gpu smoke bench goldens
