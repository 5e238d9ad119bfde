/* mto_scratch.c -- per-thread scratch memory for the oracle's temporaries (test infrastructure, see mrs_tg_oracle.h).
 *
 * The restatement follows the reference's arithmetic, not its allocation pattern (the reference allocates Eigen
 * temporaries per solve); a malloc / calloc / free per path made the all-core CPU baseline scale 7.8x on 256 threads --
 * the threads met in the allocator.  A stack of blocks per thread (__thread), grown on demand and kept for the thread's
 * life: mto_scratch_mark() / mto_scratch_release() bracket a function's temporaries. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "mrs_tg_oracle.h"

#define MTO_SCRATCH_BLOCKS 24

typedef struct {
  char* base;
  size_t cap;
} scratch_block;

static __thread scratch_block t_blocks[MTO_SCRATCH_BLOCKS];
static __thread int t_n_blocks = 0; /* allocated blocks */
static __thread int t_cur = 0;      /* block in use */
static __thread size_t t_top = 0;   /* bytes used in block t_cur */

/* a thread that ends gives its blocks back (policy worker threads, the tests' helper threads: short-lived callers) */
static pthread_key_t g_scratch_key;
static pthread_once_t g_scratch_once = PTHREAD_ONCE_INIT;
static void scratch_thread_exit(void* unused) {
  (void)unused;
  for (int i = 0; i < t_n_blocks; ++i) free(t_blocks[i].base);
  t_n_blocks = 0;
  t_cur = 0;
  t_top = 0;
}
static void scratch_make_key(void) { (void)pthread_key_create(&g_scratch_key, scratch_thread_exit); }

mto_scratch_state mto_scratch_mark(void) {
  mto_scratch_state s = {t_cur, t_top};
  return s;
}

void mto_scratch_release(mto_scratch_state s) {
  t_cur = s.block;
  t_top = s.top;
}

void* mto_scratch_alloc(size_t bytes, int zero) {
  bytes = (bytes + 63u) & ~(size_t)63u;
  if (bytes == 0) bytes = 64;
  for (;;) {
    if (t_cur < t_n_blocks) {
      if (t_top + bytes <= t_blocks[t_cur].cap) {
        void* p = t_blocks[t_cur].base + t_top;
        t_top += bytes;
        if (zero) memset(p, 0, bytes);
        return p;
      }
      ++t_cur; /* does not fit behind what the block already holds: the next block (an existing one, or a new one below) */
      t_top = 0;
      continue;
    }
    if (t_n_blocks >= MTO_SCRATCH_BLOCKS) return NULL;
    size_t cap = (size_t)1 << 20;
    while (cap < bytes) cap <<= 1;
    char* base = (char*)malloc(cap);
    if (!base) return NULL;
    if (t_n_blocks == 0) { /* first block of this thread: register the exit hook (a non-NULL value arms the destructor) */
      pthread_once(&g_scratch_once, scratch_make_key);
      (void)pthread_setspecific(g_scratch_key, (void*)&t_blocks[0]);
    }
    t_blocks[t_n_blocks].base = base;
    t_blocks[t_n_blocks].cap = cap;
    ++t_n_blocks; /* t_cur == its index, t_top == 0 */
  }
}
