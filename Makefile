# Convenience targets; the driver uses __graft_entry__.build() / pytest / bench.py directly.
.PHONY: build test-cpu test-gpu bench demo clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test-cpu: build
	python -m pytest tests -x -q -m "not gpu"
test-gpu:
	python -m pytest tests -x -q -m gpu
bench:
	python bench.py
demo: build
	g++ -std=c++17 -O2 -Iinclude examples/shaderball_demo.cpp -Lbibim_renderer_amd -lbibim_hip -Wl,-rpath,$(CURDIR)/bibim_renderer_amd -o examples/shaderball_demo
clean:
	$(MAKE) -C bibim_renderer_amd/csrc clean
	rm -f oracle/*.so oracle/_ref/*.so examples/shaderball_demo
