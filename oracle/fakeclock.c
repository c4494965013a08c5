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
 *
 * FAKECLOCK_STEP_NSEC (default 0 = frozen): every clock_gettime call advances the fake time by that many nanoseconds, the
 * way consecutive calls of a real clock differ.  With a frozen clock the reference's 101 GC-factor generators
 * (Profile.cpp:1409-1415, one default_random_engine per GC%, each seeded from now()) all get the SAME seed and draw the
 * same sequence; the statistical tests of the GC law (tests/test_gpu_histograms.py) step the clock so that the reference
 * runs the way it does in the field.  The md5 pins keep the frozen clock.
 *
 * FAKECLOCK_NO_PIN (unset by default): the reference pins worker i to CPU i (ThreadPool.cpp:32-38); a test that starts
 * several reference runs at once (the GC-law test: three runs of one work item each) would have them share CPU 0.  With
 * the variable set pthread_setaffinity_np reports success without pinning and the scheduler places the workers.  What
 * the reference computes does not depend on where its threads run.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <time.h>
#include <dlfcn.h>
#include <sys/time.h>

static long fk_sec(void)  { const char *s = getenv("FAKECLOCK_SEC");  return s ? atol(s) : 1500000000L; }
static long fk_nsec(void) { const char *s = getenv("FAKECLOCK_NSEC"); return s ? atol(s) : 123456789L; }

static long fk_step(void) { const char *s = getenv("FAKECLOCK_STEP_NSEC"); return s ? atol(s) : 0L; }

int clock_gettime(clockid_t id, struct timespec *ts) {
  static unsigned long calls = 0;
  (void)id;
  const long step = fk_step();
  const unsigned long k = step ? __atomic_fetch_add(&calls, 1UL, __ATOMIC_RELAXED) : 0UL;
  const unsigned long long ns = (unsigned long long)fk_nsec() + (unsigned long long)step * k;
  ts->tv_sec = fk_sec() + (long)(ns / 1000000000ULL);
  ts->tv_nsec = (long)(ns % 1000000000ULL);
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

int pthread_setaffinity_np(pthread_t th, size_t n, const cpu_set_t *set) {
  typedef int (*fn_t)(pthread_t, size_t, const cpu_set_t *);
  static fn_t real = 0;
  if (getenv("FAKECLOCK_NO_PIN")) return 0;
  if (!real) real = (fn_t)dlsym(RTLD_NEXT, "pthread_setaffinity_np");
  return real ? real(th, n, set) : 0;
}
