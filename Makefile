# Top-level build.  `make` builds everything that can be built where it runs:
#   jpegdec_amd/libjpegdec_amd.so   the product: host front end + HIP runtime + gfx950 kernels (hipcc)
#   oracle/liboracle.so             the checker (CPU restatement)            -- test infrastructure
#   oracle/_ref/*.so                the real reference, if /root/reference is present -- test infrastructure
#   tests/hostsim/libjda_hostsim.so wave emulator for CPU-only unit tests    -- test infrastructure
HIPCC ?= /opt/rocm/bin/hipcc
CXX   ?= g++
ARCH  ?= gfx950
CSRC  = jpegdec_amd/csrc
EXTRA ?=
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -fwrapv -pthread -Wall -Wno-unused-function -Iinclude $(EXTRA)
LIB = jpegdec_amd/libjpegdec_amd.so
LIB_SRCS = $(CSRC)/jda_frontend.cpp $(CSRC)/jda_runtime.cpp $(CSRC)/jda_pipeline.cpp $(CSRC)/jda_node.cpp $(CSRC)/jda_kernels.hip $(CSRC)/JPEGDEC.cpp
LIB_DEPS = $(LIB_SRCS) $(CSRC)/jda_runtime_internal.h $(CSRC)/jda_internal.h $(CSRC)/jda_device_core.h $(CSRC)/jda_plan.h include/jpegdec_amd.h include/JPEGDEC.h

all: lib oracle hostsim classshim

lib: $(LIB)
$(LIB): $(LIB_DEPS)
	$(HIPCC) $(HIPFLAGS) -o $@ $(LIB_SRCS)

oracle:
	$(MAKE) -C oracle all

hostsim: tests/hostsim/libjda_hostsim.so
tests/hostsim/libjda_hostsim.so: tests/hostsim/hostsim.cpp $(CSRC)/jda_frontend.cpp $(CSRC)/jda_device_core.h $(CSRC)/jda_plan.h $(CSRC)/jda_internal.h
	$(CXX) -O2 -std=c++17 -fPIC -shared -fwrapv -Wall -Wno-unused-function -Wno-unknown-pragmas -Iinclude -pthread -o $@ tests/hostsim/hostsim.cpp $(CSRC)/jda_frontend.cpp

# the reference-API driver (oracle/ref_shim.cpp) built against the product's JPEGDEC class -- test infrastructure
classshim: tests/libjpegdec_class_shim.so
tests/libjpegdec_class_shim.so: oracle/ref_shim.cpp include/JPEGDEC.h $(LIB)
	$(CXX) -O2 -std=c++17 -fPIC -shared -w -DSHIM_PRODUCT -Iinclude -o $@ oracle/ref_shim.cpp -Ljpegdec_amd -ljpegdec_amd -lpthread -Wl,-rpath,'$$ORIGIN/../jpegdec_amd'

# the same driver over the class's HOST logic alone: JPEGDEC.cpp + the host front end + a CPU stand-in for the device entry points
# (tests/class_cpu/stub_runtime.cpp: pixels from the oracle) -- test infrastructure: the recorded reference walks run on it without a
# GPU (tests/test_class_walks_cpu.py), the second build under AddressSanitizer
classcpu: tests/class_cpu/libjpegdec_class_cpu.so tests/class_cpu/walks_asan
CLASS_CPU_SRCS = oracle/ref_shim.cpp $(CSRC)/JPEGDEC.cpp $(CSRC)/jda_frontend.cpp tests/class_cpu/stub_runtime.cpp
tests/class_cpu/libjpegdec_class_cpu.so: $(CLASS_CPU_SRCS) oracle/jpegdec_oracle.c include/JPEGDEC.h include/jpegdec_amd.h
	$(CC) -O2 -std=c11 -fPIC -c -o tests/class_cpu/oracle.o oracle/jpegdec_oracle.c
	$(CXX) -O2 -std=c++17 -fPIC -shared -w -DSHIM_PRODUCT -Iinclude -o $@ $(CLASS_CPU_SRCS) tests/class_cpu/oracle.o -lpthread
tests/class_cpu/walks_asan: $(CLASS_CPU_SRCS) tests/class_cpu/walks_main.cpp oracle/jpegdec_oracle.c include/JPEGDEC.h include/jpegdec_amd.h
	$(CC) -O1 -g -std=c11 -fsanitize=address,undefined -fno-omit-frame-pointer -c -o tests/class_cpu/oracle_asan.o oracle/jpegdec_oracle.c
	$(CXX) -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -w -DSHIM_PRODUCT -Iinclude -o $@ $(CLASS_CPU_SRCS) tests/class_cpu/walks_main.cpp tests/class_cpu/oracle_asan.o -lpthread

# a plain C program on the C flavour of the API (JPEG_openFile / JPEG_decode / ...), compiled with the C compiler
cuser: tests/capi_c/c_user
tests/capi_c/c_user: tests/capi_c/c_user.c include/JPEGDEC.h $(LIB)
	$(CC) -std=c99 -O2 -Wall -Iinclude -o $@ tests/capi_c/c_user.c -Ljpegdec_amd -ljpegdec_amd -Wl,-rpath,'$$ORIGIN/../../jpegdec_amd'

# a plain C program on the node entry points (jda_node_*): one file decoded n times over every GPU of the node
nodeuser: tests/capi_c/node_user
tests/capi_c/node_user: tests/capi_c/node_user.c include/jpegdec_amd.h $(LIB)
	$(CC) -std=c99 -O2 -Wall -Iinclude -o $@ tests/capi_c/node_user.c -Ljpegdec_amd -ljpegdec_amd -Wl,-rpath,'$$ORIGIN/../../jpegdec_amd'

# the reference's jpeg_perf_test (examples/jpeg_perf_test/jpeg_perf_test.ino) as a C program on the product library: bench.py's c1 leg
perfuser: tests/capi_c/perf_user
tests/capi_c/perf_user: tests/capi_c/perf_user.c include/JPEGDEC.h $(LIB)
	$(CC) -std=c99 -O2 -Wall -Iinclude -o $@ tests/capi_c/perf_user.c -Ljpegdec_amd -ljpegdec_amd -Wl,-rpath,'$$ORIGIN/../../jpegdec_amd'

# the boundary's object semantics (class copies / moves, JPEGIMAGE without initialisation or close) -- opens only, no GPU needed
semuser: tests/capi_c/semantics_user
tests/capi_c/semantics_user: tests/capi_c/semantics_user.cpp include/JPEGDEC.h $(LIB)
	$(CXX) -std=c++17 -O2 -Wall -Iinclude -o $@ tests/capi_c/semantics_user.cpp -Ljpegdec_amd -ljpegdec_amd -Wl,-rpath,'$$ORIGIN/../../jpegdec_amd'

# the reference's own test program (MacOS/JPEGDEC_Test/JPEGDEC_Test/main.cpp) restated against the product's class
jpegtest: tests/ref_main/jpegtest_amd
tests/ref_main/jpegtest_amd: tests/ref_main/jpegtest_amd.cpp include/JPEGDEC.h $(LIB)
	$(CXX) -std=c++17 -O2 -Wall -Iinclude -o $@ tests/ref_main/jpegtest_amd.cpp -Ljpegdec_amd -ljpegdec_amd -Wl,-rpath,'$$ORIGIN/../../jpegdec_amd'

# the host front end under ASan + UBSan with a mutation driver (no GPU code in jda_frontend.cpp)
frontfuzz: tests/fuzz/frontend_fuzz
tests/fuzz/frontend_fuzz: tests/fuzz/frontend_fuzz.cpp $(CSRC)/jda_frontend.cpp $(CSRC)/jda_internal.h include/jpegdec_amd.h
	$(CXX) -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -Wall -Iinclude -pthread -o $@ tests/fuzz/frontend_fuzz.cpp $(CSRC)/jda_frontend.cpp

# the same driver under ThreadSanitizer: the interval-parallel host pre-scan (helper threads, jda_frontend.cpp) over restart streams
chunkequiv: tests/fuzz/chunk_equiv
tests/fuzz/chunk_equiv: tests/fuzz/chunk_equiv.cpp $(CSRC)/jda_frontend.cpp $(CSRC)/jda_internal.h include/jpegdec_amd.h
	$(CXX) -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -DJDA_TEST_CHUNK_BYTES_HOOK -Wall -Iinclude -pthread -o $@ tests/fuzz/chunk_equiv.cpp $(CSRC)/jda_frontend.cpp

fronttsan: tests/fuzz/frontend_tsan
tests/fuzz/frontend_tsan: tests/fuzz/frontend_fuzz.cpp $(CSRC)/jda_frontend.cpp $(CSRC)/jda_internal.h include/jpegdec_amd.h
	$(CXX) -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer -Wall -Iinclude -pthread -o $@ tests/fuzz/frontend_fuzz.cpp $(CSRC)/jda_frontend.cpp

clean:
	rm -f tests/fuzz/frontend_tsan $(LIB) tests/class_cpu/*.so tests/class_cpu/*.o tests/class_cpu/walks_asan tests/fuzz/frontend_fuzz tests/fuzz/chunk_equiv tests/ref_main/jpegtest_amd tests/hostsim/libjda_hostsim.so tests/capi_c/c_user tests/capi_c/node_user tests/capi_c/semantics_user tests/capi_c/perf_user
	$(MAKE) -C oracle clean

.PHONY: all lib oracle hostsim classshim classcpu cuser nodeuser semuser perfuser fronttsan chunkequiv jpegtest frontfuzz nodestub clean

# jda_node.cpp (host code above the C-ABI) over eight pretend devices -- test infrastructure, no GPU (tests/test_c_api.py)
nodestub: tests/node_stub/node_stub_user
tests/node_stub/node_stub_user: tests/node_stub/node_stub_user.cpp tests/node_stub/stub_pipeline.cpp $(CSRC)/jda_node.cpp include/jpegdec_amd.h
	$(CXX) -std=c++17 -O1 -g -fsanitize=thread -Wall -Iinclude -pthread -o $@ tests/node_stub/node_stub_user.cpp tests/node_stub/stub_pipeline.cpp $(CSRC)/jda_node.cpp
