/* TEST STAND-IN, not psrdada.  Declarations of exactly the multilog calls vlite-fast_amd/csrc/pb_dada_shim.c makes,
 * written from the reference's call sites (/root/reference/src/process_baseband.cu:505,519,521,98) so that the
 * shim can be COMPILED and RUN where psrdada is absent.  It pins nothing about psrdada's real ABI: argument types
 * here are what those call sites imply, no more.  Implementation: ../mock_psrdada.c. */
#ifndef MOCK_PSRDADA_MULTILOG_H
#define MOCK_PSRDADA_MULTILOG_H
#include <stdio.h>
#include <syslog.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct multilog_t multilog_t;
multilog_t *multilog_open(const char *program_name, char syslog);
int multilog_add(multilog_t *m, FILE *fptr);
int multilog(multilog_t *m, int priority, const char *format, ...);
int multilog_close(multilog_t *m);
#ifdef __cplusplus
}
#endif
#endif
