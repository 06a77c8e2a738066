// Dispatch record of the C ABI: which kernel template / tile configuration the most recent convolution entry point
// launched on the calling host thread.  The parity tests read it through yv1_last_config() to ASSERT which of the
// dispatchable configurations a given shape covered (the tile heuristics in conv.hip / wgrad.hip / conv_fp8.hip pick a
// template per shape; a test that cannot name the template it exercised proves nothing about the others).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace {
thread_local char g_cfg[512];
thread_local int g_len = 0;
}  // namespace

void yv1_cfg_reset() {
  g_len = 0;
  g_cfg[0] = 0;
}

void yv1_cfg_note(const char* fmt, ...) {
  if (g_len >= (int)sizeof(g_cfg) - 1) return;
  if (g_len) g_cfg[g_len++] = ';';
  va_list ap;
  va_start(ap, fmt);
  const int n = vsnprintf(g_cfg + g_len, sizeof(g_cfg) - g_len, fmt, ap);
  va_end(ap);
  if (n > 0) g_len = g_len + n < (int)sizeof(g_cfg) ? g_len + n : (int)sizeof(g_cfg) - 1;
}

// Copies the record (';'-separated launches of the last conv / dgrad / wgrad entry point called on this thread) into
// buf (NUL-terminated, truncated to cap) and returns its full length.
extern "C" int yv1_last_config(char* buf, int cap) {
  if (buf && cap > 0) {
    const int n = g_len < cap - 1 ? g_len : cap - 1;
    memcpy(buf, g_cfg, n);
    buf[n] = 0;
  }
  return g_len;
}
