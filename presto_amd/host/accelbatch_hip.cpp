// accelbatch_hip -- many DM-trial files through one GPU context, pipelined.
//
//   accelbatch_hip [-zmax Z] [-numharm H] [-sigma S] [-flo F] [-fhi F] [-locpow]
//                  [-noharmpolish] [-noharmremove] [-zaplist file [-baryv v]]
//                  [-rednoise] [-device D] [-quiet]  file1.fft file2.fft ... (.dat/.sdat too)
//
// Per file this is what accelsearch_hip does (the reference's accelsearch, one process
// per DM trial in the survey scripts, bin/GBNCC_search.py:360-394): search, work file,
// candidate trimming, optimisation, <root>_ACCEL_<zmax> and .cand.  Here the files move
// through three stages at once: a reader thread has the next file in memory, the GPU
// searches the current one (its upload and preparation ran beside the previous search),
// and the host finishes the previous one -- list replay, trimming, and the optimisation of
// its candidates on the device beside the running search; a writer thread writes the work
// file and the result files.  Spectra are read straight into page-locked buffers and
// uploaded from there without the host waiting.  The zapping
// and red-noise steps of the survey scripts (zapbirds -zap, rednoise, :367-377) run on
// the spectrum in device memory instead of as file-to-file passes: -zaplist applies to
// every input here (accelsearch itself zaps only a time series), -rednoise to .fft input
// (a time series is de-reddened as accelsearch does).
// In-memory search only: every trial's plane must fit the device.
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "accelgpu.h"

namespace {

void die(const char *what)
{
    fprintf(stderr, "\n%s: %s\n\n", what, accelgpu_last_error());
    exit(1);
}

// "<key text>  =  <value>" lines of a .inf file (reference src/ioinf.c:121-175)
bool inf_value(const std::string &text, const char *key, double *out)
{
    const size_t at = text.find(key);
    if (at == std::string::npos)
        return false;
    const size_t eq = text.find('=', at), nl = text.find('\n', at);
    if (eq == std::string::npos || (nl != std::string::npos && eq > nl))
        return false;
    *out = atof(text.c_str() + eq + 1);
    return true;
}

struct Options {
    int zmax = 200, numharm = 8, device = 0, norm_type = 0;
    double sigma = 2.0, flo = 1.0, fhi = 10000.0, baryv = 0.0;
    bool polish = true, harmremove = true, rednoise = false, quiet = false;
    std::string zaplist;
};

struct Trial {
    std::string path, root, inftext, error;
    bool dat_input = false;
    double N = 0, dt = 0;
    int64_t numbins = 0;
    std::vector<float> data;     // the time series (.dat / .sdat input)
    float *pinned = nullptr;     // packed spectrum (2*numbins floats) in a page-locked buffer of the pool (.fft input)
    accelgpu_setup setup{};
    int name_zmax = 0;           // -zmax as given: the reference names its files before it rounds zmax up
                                 // to an even number (src/accel_utils.c:1492-1494, then :1526)
    int64_t nopt = 0;            // candidates handed to the optimisation
    std::vector<accelgpu_cand> kept;
};

// Page-locked host buffers for spectra on their way to the GPU, all of one size (a batch's files are
// alike; a file of another size is read into ordinary memory and uploaded the blocking way).
class PinnedPool {
  public:
    float *get(size_t floats) // nullptr: not a size this pool serves
    {
        std::unique_lock<std::mutex> lk(m_);
        if (floats_ == 0) {
            floats_ = floats;
            for (int i = 0; i < 6; i++) {
                void *p = nullptr;
                if (accelgpu_host_alloc(&p, floats * sizeof(float)))
                    break;
                all_.push_back((float *) p);
                free_.push_back((float *) p);
            }
        }
        if (floats != floats_ || all_.empty())
            return nullptr;
        some_.wait(lk, [&] { return !free_.empty(); });
        float *p = free_.back();
        free_.pop_back();
        return p;
    }
    void put(float *p)
    {
        std::lock_guard<std::mutex> lk(m_);
        free_.push_back(p);
        some_.notify_one();
    }
    ~PinnedPool()
    {
        for (float *p : all_)
            (void) accelgpu_host_free(p);
    }

  private:
    std::mutex m_;
    std::condition_variable some_;
    size_t floats_ = 0;
    std::vector<float *> all_, free_;
};

// Host side of the input; no GPU call, runs on the reader thread.
std::unique_ptr<Trial> read_trial(const std::string &path, PinnedPool &pool)
{
    std::unique_ptr<Trial> t(new Trial);
    t->path = path;
    auto ends_with = [&](const char *suf) {
        const size_t n = strlen(suf);
        return path.size() > n && path.compare(path.size() - n, n, suf) == 0;
    };
    const bool is_sdat = ends_with(".sdat"), is_dat = !is_sdat && ends_with(".dat");
    t->dat_input = is_dat || is_sdat;
    if (!t->dat_input && !ends_with(".fft")) {
        t->error = "Input file ('" + path + "') must be an '.fft' or '.[s]dat' file!";
        return t;
    }
    t->root = path.substr(0, path.size() - (is_sdat ? 5 : 4));
    if (FILE *fi = fopen((t->root + ".inf").c_str(), "rb")) {
        char buf[4096];
        size_t got;
        while ((got = fread(buf, 1, sizeof(buf), fi)) > 0)
            t->inftext.append(buf, got);
        fclose(fi);
    }
    if (!inf_value(t->inftext, "Number of bins in the time series", &t->N) ||
        !inf_value(t->inftext, "Width of each time series bin", &t->dt)) {
        t->error = "Cannot read N and dt from '" + t->root + ".inf'";
        return t;
    }
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        t->error = "Cannot open '" + path + "'";
        return t;
    }
    fseek(f, 0, SEEK_END);
    const long long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    bool ok = true;
    if (is_sdat) {
        const size_t n = (size_t) (bytes / 2);
        std::vector<short> tmp(n);
        ok = fread(tmp.data(), 2, n, f) == n;
        t->data.resize(n);
        for (size_t i = 0; i < n; i++)
            t->data[i] = (float) tmp[i];
        t->numbins = (int64_t) (n / 2);
    } else {
        const size_t n = (size_t) (bytes / 4);
        float *dst = t->dat_input ? nullptr : pool.get(n);
        t->pinned = dst;
        if (!dst) {
            t->data.resize(n);
            dst = t->data.data();
        }
        ok = fread(dst, 4, n, f) == n;
        t->numbins = (int64_t) (n / 2);
    }
    fclose(f);
    if (!ok)
        t->error = "Short read on '" + path + "'";
    else if (t->dat_input && t->data.size() > 67108864)
        t->error = "The input time series '" + path + "' is too large.  Use 'realfft' first.";
    return t;
}

// Bounded hand-over from the reader thread.
class Mailbox {
  public:
    void put(std::unique_ptr<Trial> t)
    {
        std::unique_lock<std::mutex> lk(m_);
        room_.wait(lk, [&] { return q_.size() < 2; });
        q_.push_back(std::move(t));
        some_.notify_one();
    }
    std::unique_ptr<Trial> get() // nullptr: no more files
    {
        std::unique_lock<std::mutex> lk(m_);
        some_.wait(lk, [&] { return !q_.empty(); });
        std::unique_ptr<Trial> t = std::move(q_.front());
        q_.pop_front();
        room_.notify_one();
        return t;
    }

  private:
    std::mutex m_;
    std::condition_variable some_, room_;
    std::deque<std::unique_ptr<Trial>> q_;
};

struct Geometry {
    int64_t numbins = -1;
    double N = 0, dt = 0;
    bool same(const Trial &t) const { return numbins == t.numbins && N == t.N && dt == t.dt; }
};

struct Birds {
    double T = -1;
    std::vector<double> lo, hi;
    int32_t use = 0;
};

// Upload, prepare and launch one trial (beside a search still in flight).
void stage_trial(accelgpu_ctx *gpu, Trial &t, const Options &o, Geometry &geo, accelgpu_setup &su, Birds &birds)
{
    if (!geo.same(t)) { // the caller has drained the GPU
        if (accelgpu_plan(t.numbins, (long long) t.N, t.dt, o.zmax, o.numharm, o.sigma, o.flo, o.fhi, 0, &su))
            die("accelgpu_plan");
        if (accelgpu_build_kernels(gpu, su.zmax)) die("accelgpu_build_kernels");
        if (accelgpu_set_search(gpu, &su, o.norm_type, 0.f)) die("accelgpu_set_search");
        geo.numbins = t.numbins;
        geo.N = t.N;
        geo.dt = t.dt;
    }
    t.setup = su;
    t.name_zmax = o.zmax;
    if (t.dat_input) {
        if (accelgpu_load_dat(gpu, t.data.data(), (int64_t) t.data.size(), 1)) die("accelgpu_load_dat");
        // no -photon here: bin 0 becomes (1, 1) as in src/accel_utils.c:1553-1556
        if (accelgpu_set_bin0(gpu, 1.0f, 1.0f)) die("accelgpu_set_bin0");
    } else if (t.pinned) { // no host wait: the launch waits for the copy on the device
        if (accelgpu_load_fft_async(gpu, t.pinned, t.numbins)) die("accelgpu_load_fft_async");
    } else if (accelgpu_load_fft(gpu, t.data.data(), t.numbins)) {
        die("accelgpu_load_fft");
    }
    std::vector<float>().swap(t.data);
    if (!o.zaplist.empty()) {
        if (birds.T != su.T) { // the list in bins of this observation length, cut at the Nyquist bin
            int32_t nb = 0;
            birds.lo.assign(1, 0.0);
            birds.hi.assign(1, 0.0);
            int rc = accelgpu_read_birds(o.zaplist.c_str(), su.T, o.baryv, birds.lo.data(), birds.hi.data(), 0, &nb);
            if (rc == ACCELGPU_ERR_OVERFLOW) {
                birds.lo.resize((size_t) nb);
                birds.hi.resize((size_t) nb);
                rc = accelgpu_read_birds(o.zaplist.c_str(), su.T, o.baryv, birds.lo.data(), birds.hi.data(), nb, &nb);
            }
            if (rc) die("accelgpu_read_birds");
            const double hibin = (double) ((long long) t.N / 2);
            birds.use = 0;
            for (; birds.use < nb && birds.lo[(size_t) birds.use] < hibin; birds.use++)
                if (birds.hi[(size_t) birds.use] >= hibin)
                    birds.hi[(size_t) birds.use] = hibin - 1;
            birds.T = su.T;
        }
        if (accelgpu_zapbirds(gpu, birds.lo.data(), birds.hi.data(), birds.use)) die("accelgpu_zapbirds");
    }
    if (o.rednoise && !t.dat_input)
        if (accelgpu_rednoise(gpu, 6, 100, 6.0, su.T)) die("accelgpu_rednoise");
    if (accelgpu_run_launch(gpu)) die("accelgpu_run_launch");
}

// File writing off the launch path: jobs run in the order they were handed over.
class Writer {
  public:
    Writer() : th_([this] { run(); }) {}
    ~Writer() { finish(); }
    void submit(std::function<void()> job)
    {
        std::lock_guard<std::mutex> lk(m_);
        q_.push_back(std::move(job));
        some_.notify_one();
    }
    void finish()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            if (done_) return;
            done_ = true;
            some_.notify_one();
        }
        th_.join();
    }

  private:
    void run()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                some_.wait(lk, [&] { return done_ || !q_.empty(); });
                if (q_.empty())
                    return;
                job = std::move(q_.front());
                q_.pop_front();
            }
            job();
        }
    }
    std::mutex m_;
    std::condition_variable some_;
    std::deque<std::function<void()>> q_;
    bool done_ = false;
    std::thread th_;
};

// Collect a launched trial: list replay, work file (to the writer), trimming.
void collect_trial(accelgpu_ctx *gpu, Trial &t, const Options &o, PinnedPool &pool, Writer &writer)
{
    const accelgpu_setup &su = t.setup;
    int64_t nhits = 0;
    int rc;
    auto hits = std::make_shared<std::vector<accelgpu_hit>>((size_t) 1 << 18);
    while ((rc = accelgpu_run_collect(gpu, hits->data(), (int64_t) hits->size(), &nhits)) == ACCELGPU_ERR_OVERFLOW &&
           nhits > (int64_t) hits->size())
        hits->resize((size_t) nhits + 1024);
    if (rc) die("accelgpu_run_collect");
    if (t.pinned) { // the search has read its spectrum on the device: the host buffer is free
        pool.put(t.pinned);
        t.pinned = nullptr;
    }
    std::vector<accelgpu_cand> cands((size_t) nhits + 1);
    auto sig = std::make_shared<std::vector<float>>((size_t) nhits + 1);
    auto added = std::make_shared<std::vector<int32_t>>((size_t) nhits + 1);
    int64_t ncands = 0;
    if (accelgpu_replay_candidates(&su, hits->data(), nhits, cands.data(), (int64_t) cands.size(), &ncands, sig->data(),
                                   added->data()))
        die("accelgpu_replay_candidates");
    if (!t.dat_input) { // the work file of the search (src/accel_utils.c:1290-1294)
        const std::string work = t.root + "_ACCEL_" + std::to_string(t.name_zmax) + ".txtcand";
        writer.submit([su, hits, sig, added, nhits, work] {
            if (accelgpu_write_txtcand(&su, hits->data(), sig->data(), added->data(), nhits, work.c_str()))
                die("accelgpu_write_txtcand");
        });
    }
    t.nopt = 0;
    if (ncands > 0) {
        int64_t kept = 0;
        if (accelgpu_sort_and_eliminate(cands.data(), ncands, su.numharm > 1 && o.harmremove, &kept))
            die("accelgpu_sort_and_eliminate");
        t.kept.assign(cands.begin(), cands.begin() + kept);
    }
    if (!o.quiet)
        printf("%s: %lld cells above threshold, %lld candidates, %lld kept\n", t.path.c_str(), (long long) nhits,
               (long long) ncands, (long long) t.kept.size());
}

// Queue the optimisation of the trimmed list (against this trial's spectrum, beside the next
// trial's search).  One optimisation is out at a time: the one before has been collected.
void optimise_trial(accelgpu_ctx *gpu, Trial &t, const Options &o)
{
    if (t.kept.empty())
        return;
    if (accelgpu_optimize_launch(gpu, t.kept.data(), (int64_t) t.kept.size(), o.polish ? 1 : 0))
        die("accelgpu_optimize_launch");
    t.nopt = (int64_t) t.kept.size();
}

// Wait for the optimisation and hand the result files to the writer.
void write_trial(accelgpu_ctx *gpu, Trial &t, Writer &writer)
{
    if (t.nopt == 0)
        return;
    auto opt = std::make_shared<std::vector<accelgpu_optcand>>((size_t) t.nopt);
    if (accelgpu_optimize_collect(gpu, opt->data(), t.nopt)) die("accelgpu_optimize_collect");
    const std::string accelnm = t.root + "_ACCEL_" + std::to_string(t.name_zmax);
    const accelgpu_setup su = t.setup;
    const std::string inftext = t.inftext;
    writer.submit([su, opt, accelnm, inftext] {
        if (accelgpu_write_accel_files(&su, opt->data(), (int64_t) opt->size(), 0.f, accelnm.c_str(), (accelnm + ".cand").c_str(),
                                       inftext.c_str()))
            die("accelgpu_write_accel_files");
    });
}

} // namespace

int main(int argc, char **argv)
{
    Options o;
    std::vector<std::string> files;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char *flag) {
            if (i + 1 >= argc) {
                printf("\nMissing value for %s\n\n", flag);
                exit(1);
            }
            return argv[++i];
        };
        if (a == "-zmax") o.zmax = atoi(need("-zmax"));
        else if (a == "-numharm") o.numharm = atoi(need("-numharm"));
        else if (a == "-sigma") o.sigma = atof(need("-sigma"));
        else if (a == "-flo") o.flo = atof(need("-flo"));
        else if (a == "-fhi") o.fhi = atof(need("-fhi"));
        else if (a == "-device") o.device = atoi(need("-device"));
        else if (a == "-zaplist") o.zaplist = need("-zaplist");
        else if (a == "-baryv") o.baryv = atof(need("-baryv"));
        else if (a == "-ncpus") (void) need("-ncpus");
        else if (a == "-locpow") o.norm_type = 1;
        else if (a == "-median") o.norm_type = 0;
        else if (a == "-inmem" || a == "-otheropt") {}
        else if (a == "-rednoise") o.rednoise = true;
        else if (a == "-noharmpolish") o.polish = false;
        else if (a == "-noharmremove") o.harmremove = false;
        else if (a == "-quiet") o.quiet = true;
        else if (a[0] == '-') {
            printf("\nUnknown option '%s'\n\n", a.c_str());
            return 1;
        } else files.push_back(a);
    }
    if (files.empty()) {
        printf("\nusage: accelbatch_hip [search flags] file1.fft file2.fft ...\n\n");
        return 1;
    }
    if (o.zmax < 0 || o.zmax > 1200 || o.sigma < 1.0 || o.sigma > 30.0 || o.baryv < -0.1 || o.baryv > 0.1) {
        printf("\n-zmax must be in 0..1200, -sigma in 1..30 and -baryv in -0.1..0.1\n\n");
        return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    accelgpu_ctx *gpu = nullptr;
    if (accelgpu_create(&gpu, o.device)) die("accelgpu_create");

    Mailbox box;
    PinnedPool pool;
    Writer writer;
    std::thread reader([&] {
        (void) accelgpu_bind_thread(gpu); // page-locked memory is asked of the context's device
        for (const std::string &f : files)
            box.put(read_trial(f, pool));
        box.put(nullptr);
    });

    Geometry geo;
    accelgpu_setup su{};
    Birds birds;
    std::unique_ptr<Trial> launched, optimising;
    // One step for the trials behind the newest launch.  The older search is collected first:
    // the optimisation queued a step ago ran beside it on the device and is over by the time
    // its results are asked for (waiting for it before the search put its whole length, 3 ms
    // of dependent steps, on the host's path).
    auto advance = [&] {
        if (launched)
            collect_trial(gpu, *launched, o, pool, writer);
        if (optimising) {
            write_trial(gpu, *optimising, writer);
            optimising.reset();
        }
        if (launched) {
            optimise_trial(gpu, *launched, o);
            optimising = std::move(launched);
        }
    };
    size_t done = 0;
    std::chrono::steady_clock::time_point t_first; // when the first file's search was launched: set-up ends here
    for (;;) {
        std::unique_ptr<Trial> t = box.get();
        if (!t)
            break;
        if (!t->error.empty()) {
            fprintf(stderr, "\n%s\n\n", t->error.c_str());
            reader.detach();
            return 1;
        }
        if (launched && !geo.same(*t))
            advance(); // a new geometry rebuilds the plane: drain first
        std::unique_ptr<Trial> prev = std::move(launched);
        stage_trial(gpu, *t, o, geo, su, birds); // queued behind prev's search on the GPU
        launched = std::move(prev);
        advance();                               // host work and optimisation overlap t's search
        launched = std::move(t);
        if (done++ == 0)
            t_first = std::chrono::steady_clock::now();
    }
    advance();
    advance();
    reader.join();
    writer.finish();
    const auto t1 = std::chrono::steady_clock::now();
    const double secs = std::chrono::duration<double>(t1 - t0).count();
    const double steady = std::chrono::duration<double>(t1 - t_first).count();
    // the second figure leaves out what happens once per process (device and context start-up, template bank,
    // allocating and clearing the plane) and the first file's read
    printf("\n%zu files in %.3f sec (%.1f files/s; %.1f files/s from the first launch on)\n\n", done, secs, done / secs,
           done > 1 ? (done - 1) / steady : 0.0);
    accelgpu_destroy(gpu);
    return 0;
}
