# Top-level helper targets.  The product is built by __graft_entry__.build() (csrc/Makefile, host/Makefile, oracle/Makefile).
#
#   make sanitize   ASan + UBSan builds of the CPU-side pieces -- the oracle (oracle/*.c), the host k-way merge of the shard
#                   exchange (csrc/merge_host.hip, host-only compile) -- and the CPU test suites that exercise them (oracle
#                   goldens / semantics / HNSW, the gloo world-size-2/4 exchange) run under the sanitizers.  Build container
#                   only: GPU AddressSanitizer is not available on this pool.
SAN = -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -g -O1
SANDIR = build/san
ASAN_SO := $(shell gcc -print-file-name=libasan.so)
UBSAN_SO := $(shell gcc -print-file-name=libubsan.so)
CSRC = duckdb-faiss-ext_amd/csrc

.PHONY: sanitize clean-sanitize
sanitize: $(SANDIR)/liborc_san.so $(SANDIR)/san_merge
	ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $(SANDIR)/san_merge
	ORC_LIB_PATH=$(abspath $(SANDIR)/liborc_san.so) LD_PRELOAD="$(ASAN_SO) $(UBSAN_SO)" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 \
	  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4 \
	  python3 -m pytest -x -q -m "not gpu" tests/test_oracle_golden.py tests/test_oracle_semantics.py tests/test_oracle_hnsw.py \
	  tests/test_collect_bound_cpu.py tests/test_sharded_gloo.py -p no:cacheprovider
	@echo "sanitize: OK"

$(SANDIR)/liborc_san.so: oracle/orc_core.c oracle/orc_hnsw.c oracle/orc.h oracle/orc_internal.h
	@mkdir -p $(SANDIR)
	gcc $(SAN) -std=gnu11 -mavx2 -mfma -ffp-contract=off -fopenmp -fPIC -Wall -shared -o $@ oracle/orc_core.c oracle/orc_hnsw.c -lm -ldl

# merge_host.hip is plain host C++ (no HIP call): compiled as C++ with the HIP headers on the include path
$(SANDIR)/san_merge: tests/san/san_merge.cpp $(CSRC)/merge_host.hip $(CSRC)/index.h $(CSRC)/common.h
	@mkdir -p $(SANDIR)
	g++ $(SAN) -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -I$(CSRC) -x c++ $(CSRC)/merge_host.hip tests/san/san_merge.cpp -o $@ -lpthread

clean-sanitize:
	rm -rf $(SANDIR)
