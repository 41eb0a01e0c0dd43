/* The boundary of the product is a C ABI: this translation unit is plain C99 (no HIP, no C++, no torch).  It includes
 * include/frcnn_hip.h, opens libfrcnn_hip.so with dlopen and resolves every symbol named on the command line; it then
 * calls the entry points that need no GPU (version, last error, workspace sizes, argument validation).
 * Built and run by tests/test_host_logic.py::test_c_abi_from_plain_c. */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "frcnn_hip.h"

typedef int (*version_fn)(void);
typedef const char* (*last_error_fn)(void);
typedef size_t (*nms_ws_fn)(int);
typedef int (*import_plans_fn)(const int*, int);

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: abi_check libfrcnn_hip.so [symbol ...]\n");
    return 2;
  }
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 3;
  }
  int missing = 0;
  for (int i = 2; i < argc; ++i)
    if (!dlsym(lib, argv[i])) {
      fprintf(stderr, "missing symbol %s\n", argv[i]);
      ++missing;
    }
  if (missing) return 4;
  /* POSIX idiom: dlsym returns an object pointer, ISO C has no cast from it to a function pointer */
  version_fn version;
  last_error_fn last_error;
  nms_ws_fn nms_ws;
  import_plans_fn import_plans;
  *(void**)(&version) = dlsym(lib, "frcnn_version");
  *(void**)(&last_error) = dlsym(lib, "frcnn_last_error");
  *(void**)(&nms_ws) = dlsym(lib, "frcnn_nms_ws_bytes");
  *(void**)(&import_plans) = dlsym(lib, "frcnn_conv2d_import_plans");
  if (!version || !last_error || !nms_ws || !import_plans) return 4;
  if (version() < 1) return 5;
  if (nms_ws(6000) < (size_t)6000 * 94 * 8) return 6;               /* the suppression bit-matrix must fit */
  int bad[13] = {1, 8, 8, 4, 4, 1, 1, 1, 0, 1, /* tile */ 99, 1, 1};
  if (import_plans(bad, 1) == FRCNN_OK) return 7;                     /* argument validation without a GPU */
  if (!last_error() || !strstr(last_error(), "not a valid plan")) return 8;
  printf("abi ok: version %d, %d symbols, nms workspace %zu bytes\n", version(), argc - 2, nms_ws(6000));
  return 0;
}
