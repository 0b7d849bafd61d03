# Builds the three native libraries in-tree:
#   raytracer_amd/librt_mi355x.so  HIP kernels + C-ABI (include/rt_abi.h), gfx950 only
#   raytracer_amd/librt_host.so    CPU scene layer     (include/rt_host.h)
#   oracle/liboracle.so            CPU golden model, test infrastructure only
HIPCC ?= hipcc
CXX ?= g++
CSRC := raytracer_amd/csrc
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
            -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function
HIP_SRCS := $(CSRC)/rt_api.hip $(CSRC)/rt_kernels.hip $(CSRC)/rt_bvh.hip $(CSRC)/rt_sort.hip $(CSRC)/rt_ploc.hip $(CSRC)/rt_sah.hip $(CSRC)/rt_bluenoise.hip
HIP_OBJS := $(HIP_SRCS:.hip=.o)
HIP_HDRS := $(wildcard $(CSRC)/*.h) include/rt_abi.h

all: raytracer_amd/librt_mi355x.so raytracer_amd/librt_host.so oracle/liboracle.so raytracer_amd/rt_render tools/valu_issue_bench tools/valu_class_bench tools/valu_exec_bench

$(CSRC)/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

raytracer_amd/librt_mi355x.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(HIP_OBJS)

raytracer_amd/librt_host.so: raytracer_amd/host/rt_host.cpp raytracer_amd/host/rt_scene.cpp include/rt_host.h include/rt_abi.h
	$(CXX) -O2 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra -shared -o $@ raytracer_amd/host/rt_host.cpp raytracer_amd/host/rt_scene.cpp

# headless frame loop (main.cpp replacement) on top of the two libraries
raytracer_amd/rt_render: raytracer_amd/host/rt_render.cpp raytracer_amd/host/raytracing.hpp raytracer_amd/librt_mi355x.so raytracer_amd/librt_host.so
	$(CXX) -O2 -std=c++17 -Wall -o $@ raytracer_amd/host/rt_render.cpp -Lraytracer_amd -lrt_mi355x -lrt_host '-Wl,-rpath,$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

# microbenchmark behind bench.py's roofline peak (profiles/r02_valu_issue.txt)
tools/valu_issue_bench: tools/valu_issue_bench.hip
	$(HIPCC) --offload-arch=gfx950 -O2 -o $@ $<
# which instruction kinds issue at the fast rate, and whether the exec mask matters (profiles/r03_valu_classes.txt, r03_valu_exec_mask.txt)
tools/valu_class_bench: tools/valu_class_bench.hip
	$(HIPCC) --offload-arch=gfx950 -O2 -o $@ $<
tools/valu_exec_bench: tools/valu_exec_bench.hip
	$(HIPCC) --offload-arch=gfx950 -O2 -o $@ $<

oracle/liboracle.so: oracle/rt_oracle.cpp include/rt_abi.h
	$(MAKE) -C oracle

clean:
	rm -f $(HIP_OBJS) raytracer_amd/rt_render raytracer_amd/librt_mi355x.so raytracer_amd/librt_host.so oracle/liboracle.so

.PHONY: all clean
