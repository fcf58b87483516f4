# Convenience targets (the driver uses __graft_entry__.py, bench.py and pytest directly).
.PHONY: build test test-gpu bench consts golden clean
build:
	python __graft_entry__.py
test:
	python -m pytest tests -x -q -m "not gpu"
test-gpu:
	python -m pytest tests -x -q -m gpu
bench:
	python bench.py
consts:
	python tools/gen_consts.py
golden:
	python tests/golden/gen_golden.py
clean:
	$(MAKE) -C plonky2-bls12-381-pairing_amd clean
	$(MAKE) -C oracle clean
	rm -f tests/hostsim/libhostsim.so
