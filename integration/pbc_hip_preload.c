/* pbc_hip_preload.c -- the drop-in for UNMODIFIED programs.
 *
 * Every PBC program builds its pairing through pairing_init_pbc_param (ecc/pairing.c:74-86; pairing_init_set_buf /
 * pairing_init_set_str, :88-102, end there), which installs the map / prod_pairings / pp_* function pointers of the
 * parameter's type.  This file interposes that one function: it runs the stock initialisation, writes the parameter
 * back to text with the reference's own pbc_param_out_str (include/pbc_param.h:38) and calls pbc_hip_attach
 * (pbc_hip_glue.c), which swaps the pointers for the GPU-backed ones.  No source change in the program.
 *
 *   shared libpbc:   LD_PRELOAD=libpbc_hip_preload.so PBC_HIP_LIB=/path/libpbc_hip.so ./bls < a.param
 *                    (built by `make -C oracle preload`: this file + pbc_hip_glue.c, -shared)
 *   static libpbc:   cc prog.o pbc_hip_preload.c pbc_hip_glue.c -DPBC_HIP_LINK_WRAP \
 *                       -Wl,--wrap=pairing_init_set_buf,--wrap=pairing_init_set_str,--wrap=pairing_init_pbc_param \
 *                       libpbc.a -lgmp -ldl -lpthread
 *                    (ld's --wrap redirects references BETWEEN object files only: the call from pairing_init_set_buf
 *                    to pairing_init_pbc_param sits inside ecc/pairing.o, so the two entry points a program calls are
 *                    wrapped as well; each pairing is attached exactly once)
 *
 * There is no CPU path behind it: if the GPU object cannot be built (no device, a pairing type the engine does not
 * have), the program ends with PBC's own pbc_die, as it would for a bad parameter file.  PBC_HIP_VERBOSE=1 prints at exit
 * how many calls ran on the GPU.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include "pbc_hip_glue.h"

static void attach_or_die(struct pairing_s *pairing, pbc_param_ptr p) {
  char *text = NULL;
  size_t len = 0;
  FILE *f = open_memstream(&text, &len);
  if (!f) pbc_die("pbc_hip preload: open_memstream failed");
  pbc_param_out_str(f, p);
  fclose(f);
  if (pbc_hip_attach(pairing, text, len)) pbc_die("pbc_hip preload: cannot put this pairing on the GPU (see the message above)");
  free(text);
}

#ifdef PBC_HIP_LINK_WRAP
#include <string.h>
void __real_pairing_init_pbc_param(struct pairing_s *pairing, pbc_param_ptr p);
int __real_pairing_init_set_buf(struct pairing_s *pairing, const char *s, size_t len);
int __real_pairing_init_set_str(struct pairing_s *pairing, const char *s);
void __wrap_pairing_init_pbc_param(struct pairing_s *pairing, pbc_param_ptr p) {     /* programs that call it themselves */
  __real_pairing_init_pbc_param(pairing, p);
  attach_or_die(pairing, p);
}
int __wrap_pairing_init_set_buf(struct pairing_s *pairing, const char *s, size_t len) {
  int rc = __real_pairing_init_set_buf(pairing, s, len);
  if (!rc && pbc_hip_attach(pairing, s, len)) pbc_die("pbc_hip: cannot put this pairing on the GPU (see the message above)");
  return rc;
}
int __wrap_pairing_init_set_str(struct pairing_s *pairing, const char *s) {
  int rc = __real_pairing_init_set_str(pairing, s);
  if (!rc && pbc_hip_attach(pairing, s, strlen(s))) pbc_die("pbc_hip: cannot put this pairing on the GPU (see the message above)");
  return rc;
}
#else
void pairing_init_pbc_param(struct pairing_s *pairing, pbc_param_ptr p) {
  static void (*real)(struct pairing_s *, pbc_param_ptr);
  if (!real) *(void **) &real = dlsym(RTLD_NEXT, "pairing_init_pbc_param");
  if (!real) pbc_die("pbc_hip preload: the stock pairing_init_pbc_param is not visible (is libpbc linked dynamically?)");
  real(pairing, p);
  attach_or_die(pairing, p);
}
#endif
