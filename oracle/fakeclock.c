/* oracle/fakeclock.c -- TEST INFRASTRUCTURE ONLY.
 *
 * LD_PRELOAD shim that freezes the wall clock seen by the *unmodified* reference binary
 * (oracle/_ref/simuReads).  The reference has no seed option: every generator is seeded
 * from the clock --
 *   lib/threadpool/ThreadPool.cpp:41  mt19937(chrono::system_clock::now())   (worker RNG)
 *   lib/profile/Profile.cpp:1410      default_random_engine(system_clock)   (GC factors)
 *   lib/genome/Genome.cpp:852         srand(time(0))                        (haplotype picks)
 * With FAKECLOCK_SEC / FAKECLOCK_NSEC set and `threads = 1` its FASTQ output becomes a pure
 * function of (inputs, fake time), which is what the oracle restatement is pinned against.
 * Nothing of the reference is replaced: only the time source is pinned.
 */
#define _GNU_SOURCE
#include <stdlib.h>
#include <time.h>
#include <sys/time.h>

static long fk_sec(void)  { const char *s = getenv("FAKECLOCK_SEC");  return s ? atol(s) : 1500000000L; }
static long fk_nsec(void) { const char *s = getenv("FAKECLOCK_NSEC"); return s ? atol(s) : 123456789L; }

int clock_gettime(clockid_t id, struct timespec *ts) {
  (void)id;
  ts->tv_sec = fk_sec();
  ts->tv_nsec = fk_nsec();
  return 0;
}
time_t time(time_t *t) {
  time_t v = (time_t)fk_sec();
  if (t) *t = v;
  return v;
}
int gettimeofday(struct timeval *tv, void *tz) {
  (void)tz;
  if (tv) { tv->tv_sec = fk_sec(); tv->tv_usec = fk_nsec() / 1000; }
  return 0;
}
