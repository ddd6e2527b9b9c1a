// Vectorised environment worker pool (host side of the rollout path).
//
// The reference steps ONE Python env inline per learner update (DQN.py:316) and pays an H2D +
// D2H round trip per step in select_action (DQN.py:77,83).  Here n independent env instances are
// stepped by persistent host worker threads straight into pinned staging buffers, so one
// hipMemcpyAsync moves a whole observation batch and one act launch serves every instance.
// Dynamics restate the same published equations as freerl_amd/envs.py (Pendulum-v1, CartPole-v1)
// plus the synthetic linear-Gaussian task; plain C++, no HIP in this header.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace frl {

enum EnvKind : int { ENV_PENDULUM = 0, ENV_CARTPOLE = 1, ENV_SYNLINEAR = 2, ENV_SYNLINEAR_DISCRETE = 3, ENV_PENDULUM_SHORT = 4,
                     ENV_SYNBAND_WIDE = 5,     // Humanoid-v4's dims (obs 376, act 17) on banded linear dynamics: BASELINE config 4's rollout leg
                     ENV_CALLBACK = 100 };     // caller-supplied environments behind one vectorised step / reset callback

typedef int (*EnvStepFn)(void* user, const float* actions, float* next_obs, float* reward, uint8_t* terminated, uint8_t* truncated,
                         float* obs_next);
typedef int (*EnvResetFn)(void* user, float* obs_out);

struct XorShift {                       // xoshiro256** seeded by splitmix64
    uint64_t s[4];
    static uint64_t sm(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    void seed(uint64_t x) { for (auto& v : s) v = sm(x); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (next() >> 11) * (1.0 / 9007199254740992.0); }          // [0,1)
    double uniform(double lo, double hi) { return lo + (hi - lo) * uniform(); }
    double normal() {
        const double u = 1.0 - uniform(), v = uniform();
        return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v);
    }
};

struct EnvSpec {
    int kind, obs_dim, act_dim, discrete, n_actions, max_steps, state_dim;
    float max_action;
};

inline EnvSpec env_spec(int kind) {
    switch (kind) {
        case ENV_PENDULUM: return {kind, 3, 1, 0, 0, 200, 2, 2.f};
        case ENV_PENDULUM_SHORT: return {kind, 3, 1, 0, 0, 40, 2, 2.f};
        case ENV_CARTPOLE: return {kind, 4, 1, 1, 2, 500, 4, 0.f};
        case ENV_SYNLINEAR: return {kind, 8, 2, 0, 0, 200, 8, 1.f};
        case ENV_SYNLINEAR_DISCRETE: return {kind, 8, 1, 1, 4, 200, 8, 0.f};
        case ENV_SYNBAND_WIDE: return {kind, 376, 17, 0, 0, 1000, 376, 0.4f};
    }
    return {-1, 0, 0, 0, 0, 0, 0, 0.f};
}

class EnvPool {
public:
    EnvSpec spec;
    int n;
    std::vector<double> state;          // [n][state_dim]
    std::vector<int> t;                 // steps in the current episode
    std::vector<double> ep_return;
    std::vector<long long> ep_count;    // finished episodes per env (a learner's exploration decay counts ITS episodes, TD3.py:425-427)
    std::vector<XorShift> rng;
    EnvStepFn cb_step = nullptr;        // ENV_CALLBACK
    EnvResetFn cb_reset = nullptr;
    void* cb_user = nullptr;
    int cb_error = 0;                   // last non-zero return of a callback
    std::vector<double> A, B;           // SynLinear: A[O][O], B[O][2]
    // episode statistics since the last read
    std::atomic<long long> episodes{0};
    double return_sum = 0.0;
    std::mutex stat_mu;

    EnvPool(int kind, int n_envs, int n_threads, uint64_t seed, const double* params, int n_params)
        : spec(env_spec(kind)), n(n_envs) {
        state.assign((size_t)n * spec.state_dim, 0.0);
        t.assign(n, 0);
        ep_return.assign(n, 0.0);
        ep_count.assign(n, 0);
        rng.resize(n);
        for (int i = 0; i < n; ++i) rng[i].seed(seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull * (uint64_t)(i + 1));
        if (kind == ENV_SYNLINEAR || kind == ENV_SYNLINEAR_DISCRETE) {
            const int O = spec.obs_dim;
            A.assign((size_t)O * O, 0.0);
            B.assign((size_t)O * 2, 0.0);
            if (params && n_params == O * O + O * 2) {
                std::memcpy(A.data(), params, sizeof(double) * O * O);
                std::memcpy(B.data(), params + O * O, sizeof(double) * O * 2);
            } else {                    // default: a damped rotation + fixed input map
                for (int i = 0; i < O; ++i) A[(size_t)i * O + (i + 1) % O] = 0.95;
                for (int i = 0; i < O; ++i) { B[i * 2] = 0.5 * std::sin(i + 1.0); B[i * 2 + 1] = 0.5 * std::cos(2.0 * i); }
            }
        }
        n_workers = n_threads < 1 ? 1 : n_threads;
        if (n_workers > 1)
            for (int w = 1; w < n_workers; ++w) workers.emplace_back([this, w] { worker_loop(w); });
    }

    // caller-supplied environments: the callbacks step / reset all n of them (gymnasium protocol on the other side)
    EnvPool(int n_envs, int obs_dim, int act_dim, int n_actions, float max_action, EnvStepFn step_fn, EnvResetFn reset_fn, void* user)
        : spec{ENV_CALLBACK, obs_dim, act_dim, n_actions > 0 ? 1 : 0, n_actions, 0, 0, max_action}, n(n_envs), cb_step(step_fn),
          cb_reset(reset_fn), cb_user(user) {
        t.assign(n, 0);
        ep_return.assign(n, 0.0);
        ep_count.assign(n, 0);
    }

    ~EnvPool() {
        quit.store(true);
        generation.fetch_add(1);
        { std::lock_guard<std::mutex> lk(mu); }
        cv.notify_all();
        for (auto& th : workers) th.join();
    }

    void reset_one(int i) {
        double* s = &state[(size_t)i * spec.state_dim];
        XorShift& g = rng[i];
        switch (spec.kind) {
            case ENV_PENDULUM: case ENV_PENDULUM_SHORT:
                s[0] = g.uniform(-M_PI, M_PI); s[1] = g.uniform(-1.0, 1.0); break;
            case ENV_CARTPOLE:
                for (int k = 0; k < 4; ++k) s[k] = g.uniform(-0.05, 0.05); break;
            default:
                for (int k = 0; k < spec.state_dim; ++k) s[k] = g.normal();
        }
        t[i] = 0;
        ep_return[i] = 0.0;
    }

    void observe(int i, float* obs) const {
        const double* s = &state[(size_t)i * spec.state_dim];
        if (spec.kind == ENV_PENDULUM || spec.kind == ENV_PENDULUM_SHORT) {
            obs[0] = (float)std::cos(s[0]); obs[1] = (float)std::sin(s[0]); obs[2] = (float)s[1];
        } else {
            for (int k = 0; k < spec.obs_dim; ++k) obs[k] = (float)s[k];
        }
    }

    // one transition of env i; `a` = env-unit action (continuous) or action index (discrete)
    void step_one(int i, const float* a, float* reward, uint8_t* terminated, uint8_t* truncated) {
        double* s = &state[(size_t)i * spec.state_dim];
        double r = 0.0;
        bool term = false;
        switch (spec.kind) {
            case ENV_PENDULUM: case ENV_PENDULUM_SHORT: {
                const double g = 10.0, m = 1.0, l = 1.0, dt = 0.05;
                double u = a[0]; u = u < -2.0 ? -2.0 : (u > 2.0 ? 2.0 : u);
                double th = s[0], thdot = s[1];
                double ang = std::fmod(th + M_PI, 2 * M_PI);
                if (ang < 0) ang += 2 * M_PI;
                ang -= M_PI;
                r = -(ang * ang + 0.1 * thdot * thdot + 0.001 * u * u);
                thdot = thdot + (3 * g / (2 * l) * std::sin(th) + 3.0 / (m * l * l) * u) * dt;
                thdot = thdot < -8.0 ? -8.0 : (thdot > 8.0 ? 8.0 : thdot);
                s[0] = th + thdot * dt; s[1] = thdot;
                break;
            }
            case ENV_CARTPOLE: {
                const double gravity = 9.8, masscart = 1.0, masspole = 0.1, length = 0.5, tau = 0.02;
                const double force = ((int)a[0] == 1) ? 10.0 : -10.0, total = masspole + masscart, pml = masspole * length;
                double x = s[0], xd = s[1], th = s[2], thd = s[3];
                const double c = std::cos(th), sn = std::sin(th);
                const double temp = (force + pml * thd * thd * sn) / total;
                const double thacc = (gravity * sn - c * temp) / (length * (4.0 / 3.0 - masspole * c * c / total));
                const double xacc = temp - pml * thacc * c / total;
                x += tau * xd; xd += tau * xacc; th += tau * thd; thd += tau * thacc;
                s[0] = x; s[1] = xd; s[2] = th; s[3] = thd;
                const double th_thr = 12 * 2 * M_PI / 360;
                term = x < -2.4 || x > 2.4 || th < -th_thr || th > th_thr;
                r = 1.0;
                break;
            }
            case ENV_SYNBAND_WIDE: {
                // s'[r] = 0.6 s[r] + 0.25 s[r-1] + 0.1 s[r+1] + 0.5 a[r mod 17] + 0.05 eps: three neighbours per row instead of a
                // 376 x 376 matrix, so that 10^4..10^5 instances step in the time real MuJoCo instances would need worker processes
                const int O = spec.obs_dim, A = spec.act_dim;
                double av[32];
                for (int k = 0; k < A; ++k) { const double x = a[k] / 0.4; av[k] = x < -1.0 ? -1.0 : (x > 1.0 ? 1.0 : x); }
                double prev = s[O - 1], first = s[0], sq = 0.0, mx = 0.0, aa = 0.0;
                for (int r_ = 0; r_ < O; ++r_) {
                    const double cur = s[r_], nxt = r_ + 1 < O ? s[r_ + 1] : first;
                    const double v = 0.6 * cur + 0.25 * prev + 0.1 * nxt + 0.5 * av[r_ % A] + 0.05 * rng[i].normal();
                    prev = cur;
                    s[r_] = v;
                    sq += v * v;
                    mx = std::fmax(mx, std::fabs(v));
                }
                for (int k = 0; k < A; ++k) aa += av[k] * av[k];
                r = -(sq / O + 0.1 * aa / A);
                term = mx > 8.0;
                break;
            }
            default: {
                const int O = spec.obs_dim;
                double av[2] = {0.0, 0.0};
                if (spec.discrete) {
                    const int k = (int)a[0];
                    av[k / 2] = (k % 2 == 0) ? 1.0 : -1.0;
                } else {
                    for (int k = 0; k < 2; ++k) av[k] = a[k] < -1.f ? -1.0 : (a[k] > 1.f ? 1.0 : (double)a[k]);
                }
                double nx[16];
                double sq = 0.0, mx = 0.0;
                for (int r_ = 0; r_ < O; ++r_) {
                    double acc = B[r_ * 2] * av[0] + B[r_ * 2 + 1] * av[1] + 0.05 * rng[i].normal();
                    for (int c_ = 0; c_ < O; ++c_) acc += A[(size_t)r_ * O + c_] * s[c_];
                    nx[r_] = acc;
                    sq += acc * acc;
                    mx = std::fmax(mx, std::fabs(acc));
                }
                for (int r_ = 0; r_ < O; ++r_) s[r_] = nx[r_];
                r = -(sq + 0.1 * (av[0] * av[0] + av[1] * av[1])) / O;
                term = mx > 6.0;
            }
        }
        t[i] += 1;
        ep_return[i] += r;
        *reward = (float)r;
        *terminated = term ? 1 : 0;
        *truncated = (t[i] >= spec.max_steps) ? 1 : 0;
    }

    // Vector step: actions [n][act_dim] -> next_obs (terminal observation of the transition),
    // reward, terminated, truncated, and `obs_next` = what the policy sees next (the reset
    // observation when the episode ended: the reference resets inline, DQN.py:323-335).
    void step(const float* actions, float* next_obs, float* reward, uint8_t* term, uint8_t* trunc, float* obs_next) {
        if (spec.kind == ENV_CALLBACK) {
            cb_error = 0;                                   // a failed callback is reported once, by the call that saw it
            const int rc = cb_step(cb_user, actions, next_obs, reward, term, trunc, obs_next);
            if (rc) { cb_error = rc; return; }
            long long eps = 0;
            double ret = 0.0;
            for (int i = 0; i < n; ++i) {
                t[i] += 1;
                ep_return[i] += reward[i];
                if (term[i] || trunc[i]) { ++eps; ret += ep_return[i]; ++ep_count[i]; t[i] = 0; ep_return[i] = 0.0; }
            }
            if (eps) { std::lock_guard<std::mutex> lk(stat_mu); episodes += eps; return_sum += ret; }
            return;
        }
        job = Job{actions, next_obs, reward, term, trunc, obs_next};
        run_parallel();
    }

    void reset_all(float* obs_out) {
        if (spec.kind == ENV_CALLBACK) {
            const int rc = cb_reset(cb_user, obs_out);
            cb_error = rc;
            for (int i = 0; i < n; ++i) { t[i] = 0; ep_return[i] = 0.0; }
            return;
        }
        for (int i = 0; i < n; ++i) { reset_one(i); observe(i, obs_out + (size_t)i * spec.obs_dim); }
    }

private:
    struct Job { const float* actions; float* next_obs; float* reward; uint8_t* term; uint8_t* trunc; float* obs_next; } job{};
    // Worker hand-off without a futex on the hot path: a rollout loop calls step() every 0.1 - 1 ms, and waking threads
    // through a condition variable cost 25 - 40 us per call (more than stepping 512 of the built-in envs).  Workers poll
    // `generation` for kSpinUs after their last job and only then sleep on the condition variable; the caller polls `pending`.
    // Pools too small to amortise even that (under kEnvsPerWorker envs per worker) use fewer workers, down to the caller alone.
    static constexpr int kEnvsPerWorker = 64;
    static constexpr int kSpinUs = 2000;
    int n_workers = 1, active = 1;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<long long> generation{0};
    std::atomic<int> pending{0}, sleepers{0};
    std::atomic<bool> quit{false};

    static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }

    void do_range(int w) {
        const int per = (n + active - 1) / active, lo = w * per, hi = lo + per > n ? n : lo + per;
        const int O = spec.obs_dim, A_ = spec.act_dim;
        long long eps = 0;
        double ret = 0.0;
        for (int i = lo; i < hi; ++i) {
            step_one(i, job.actions + (size_t)i * A_, job.reward + i, job.term + i, job.trunc + i);
            observe(i, job.next_obs + (size_t)i * O);
            if (job.term[i] || job.trunc[i]) {
                ++eps;
                ret += ep_return[i];
                ++ep_count[i];
                reset_one(i);
                observe(i, job.obs_next + (size_t)i * O);
            } else {
                std::memcpy(job.obs_next + (size_t)i * O, job.next_obs + (size_t)i * O, sizeof(float) * O);
            }
        }
        if (eps) {
            std::lock_guard<std::mutex> lk(stat_mu);
            episodes += eps;
            return_sum += ret;
        }
    }

    void run_parallel() {
        active = std::min(n_workers, std::max(1, (n + kEnvsPerWorker - 1) / kEnvsPerWorker));
        if (active == 1) { do_range(0); return; }
        pending.store(n_workers - 1, std::memory_order_relaxed);       // every worker acknowledges every job (idle ones at once),
                                                                       // so `active` is never read across two generations
        generation.fetch_add(1);                         // (seq_cst: publishes job / active / pending; ordered against `sleepers`)
        if (sleepers.load() > 0) {
            { std::lock_guard<std::mutex> lk(mu); }      // a worker between its predicate check and its wait holds mu
            cv.notify_all();
        }
        do_range(0);
        while (pending.load(std::memory_order_acquire) != 0) cpu_relax();
    }

    void worker_loop(int w) {
        long long seen = 0;
        for (;;) {
            const auto t_idle = std::chrono::steady_clock::now();
            for (int spins = 0; generation.load(std::memory_order_acquire) == seen; ++spins) {
                cpu_relax();
                if ((spins & 255) == 255 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_idle).count() > kSpinUs) {
                    std::unique_lock<std::mutex> lk(mu);
                    sleepers.fetch_add(1);
                    cv.wait(lk, [&] { return generation.load() != seen; });
                    sleepers.fetch_sub(1);
                    break;
                }
            }
            seen = generation.load(std::memory_order_acquire);
            if (quit.load()) return;
            if (w < active) do_range(w);
            pending.fetch_sub(1, std::memory_order_release);
        }
    }
};

}  // namespace frl
