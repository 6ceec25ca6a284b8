// The library's source hash (include/qwen3_hip.h: q3_build_id).  Kept in its own translation unit: the Makefile
// rebuilds this object whenever the hash of csrc/* + the header changes, so the id can never lag behind the code.
#ifndef Q3_BUILD_ID
#define Q3_BUILD_ID "unknown"
#endif
extern "C" const char* q3_build_id(void) { return Q3_BUILD_ID; }
