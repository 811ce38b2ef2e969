# oracle/_ref: the one file of the reference's path that compiles in this image as it lies, with no stand-ins:
#   /root/reference/src/util.c   send_string / send_int / send_double (:51-89, the SIGPROC header encoder that
#                                write_sigproc_header, src/process_baseband.cu:226-270, is made of) and
#                                check_name / check_id / check_coords (:91-152, the `-w 1` source selection)
# It is C++ (util.h uses default arguments) and needs only libc / libm.  Built in the build container only
# (/root/reference does not exist on the GPU box); the output stays out of git history (.gitignore) but travels
# with the snapshot.  Used by tests/golden/make_util_golden.py to generate tests/golden/reference_util.json, and by
# tests/test_ref_util.py (when present) to re-check the committed fixtures against the reference itself.
# Test infrastructure: nothing in the product links or loads it.
REF ?= /root/reference/src
CXX ?= g++

_ref/libref_util.so: $(REF)/util.c $(REF)/util.h
	mkdir -p _ref
	$(CXX) -O1 -shared -fPIC -I$(REF) -o $@ $(REF)/util.c -lm
