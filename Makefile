# Convenience targets; the driver itself goes through __graft_entry__.build() / pytest / bench.py.
.PHONY: all lib oracle test test-gpu example clean
all: lib oracle
lib:
	$(MAKE) -C butterfly_amd/csrc
	$(MAKE) -C butterfly_amd/csrc experimental
oracle:
	$(MAKE) -C oracle
test: all
	python -m pytest tests -x -q -m "not gpu"
test-gpu: all
	python -m pytest tests -x -q -m gpu
example: lib
	gcc -O2 -std=gnu11 -Iinclude examples/helm2_bie_device.c -Lbutterfly_amd/csrc -lbfhip -lm -Wl,-rpath,$(CURDIR)/butterfly_amd/csrc -o examples/helm2_bie_device
	gcc -O2 -std=gnu11 -Iinclude examples/sharded_apply.c -Lbutterfly_amd/csrc -lbfhip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$(CURDIR)/butterfly_amd/csrc -Wl,-rpath,/opt/rocm/lib -o examples/sharded_apply
clean:
	$(MAKE) -C butterfly_amd/csrc clean
	$(MAKE) -C oracle clean
	rm -f examples/helm2_bie_device examples/sharded_apply
