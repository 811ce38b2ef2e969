// process_baseband: the reference's executable as a native host program above the C ABI of include/pb_hip.h
// (no HIP headers here: device work, streams and page-locked memory all sit behind libpb_hip.so).
//
// Drop-in under scripts/start_process:50 of the reference:
//     process_baseband -k <bb_key> -K <fb_key> -w <0|1|2> -b <2|4|8> -g <gpu> -o -C <coadd_key>
// (+ -P 1|2  -r 0|1|2  -i  -s  -t; the legacy `-p N` of scripts/baseband_test:26 is accepted and ignored, as
// the reference's getopt string silently does).  Mirrors /root/reference/src/process_baseband.cu main():
//   :358-470   option parsing (keys are hexadecimal: `-k 40` means 0x40)
//   :784-1000  per-observation set-up: ring header, first frame, file names, SIGPROC / psrdada headers
//   :1015-1496 per-second loop: frames are placed by their own headers, a second is dispatched only once a
//              frame of the NEXT second has been seen, so the last second of an observation is dropped
//   :1108-1458 ten 100-ms segments per second -> ONE pb_submit_vdif_at + pb_process call
//   :1416-1441 coadd ring (one write per segment), .fil / _kur.fil;  :1482-1494 output ring (10 s, then 1 s)
//   :1538-1556 the PROFILE lines (-t)
// Same behaviour, log lines and bytes as vlite-fast_amd/process_baseband.py (tests compare the two).
// Extensions: --replay FILE.. (dump files instead of ring -k), --out-sink / --co-sink FILE (instead of rings
// -K / -C), --datadir, --logdir, --no-control, --fft-backend lds|hipfft, --taps 1|4, --rows-per-seg N,
// --dump-headers (print the two headers of the first observation as hex and exit; no GPU needed),
// --control-port N (test hook: commands on UDP 127.0.0.1:N instead of the multicast group).
// Rings: the psrdada shim of include/pb_dada.h is loaded at run time (dlopen of $PB_DADA_LIB or
// libpb_dada.so beside this program), so the program builds and runs replays where psrdada is absent.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <getopt.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "pb_dada.h"
#include "pb_hip.h"

namespace {

constexpr int SEG_PER_SEC = 10;
constexpr size_t VD_FRM = PB_VDIF_FRAME, VD_DAT = PB_VDIF_DATA;
constexpr size_t DADA_HDR = 4096;
const char *MC_GROUP = "224.3.29.71";   // src/multicast.h:14
constexpr int MC_READER_PORT = 20000;   // src/multicast.h:16
constexpr char CMD_QUIT = 'Q';          // src/def.h:6
const char *DEF_LOGDIR = "/home/vlite-master/mtk/logs";   // src/def.h:26
const char *DEF_DATADIR = "/mnt/ssd/fildata";             // src/def.h:28
// `-w 1` writes only for the sources a site file lists (site policy, src/util.c:91-152 of the reference; out of the
// hot path's scope, so it is data, not code): $PB_WRITE_ALLOW, else <dir of this program>/../site/write_allow.txt,
// lines `name <substring of NAME>` / `dataid <substring of DATAID>`; no file: -w 1 writes nothing
struct WriteAllow {
    std::vector<std::string> names, ids;
    std::vector<std::pair<double, double>> coords;     // `coords <ra> <dec>` [rad], check_coords (src/util.c:136-152)
    std::string path;
    bool loaded = false, ok = false;
    void load()
    {
        if (loaded) return;
        loaded = true;
        if (const char *e = getenv("PB_WRITE_ALLOW")) path = e;
        else {
            char self[4096];
            ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
            if (n <= 0) return;
            self[n] = 0;
            std::string d(self);
            path = d.substr(0, d.rfind('/')) + "/../site/write_allow.txt";
        }
        FILE *f = fopen(path.c_str(), "r");
        if (!f) return;
        ok = true;
        char line[512];
        while (fgets(line, sizeof line, f)) {
            if (char *h = strchr(line, '#')) *h = 0;
            char key[32], val[256];
            double ra, de;
            if (sscanf(line, "%31s %lf %lf", key, &ra, &de) == 3 && !strcmp(key, "coords")) { coords.push_back({ra, de}); continue; }
            if (sscanf(line, "%31s %255s", key, val) != 2) continue;
            if (!strcmp(key, "name")) names.push_back(val);
            else if (!strcmp(key, "dataid")) ids.push_back(val);
        }
        fclose(f);
    }
} g_allow;

double now()
{
    timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
}

struct Args {
    unsigned key_in = 0x40, key_out = 0, key_co = 0;
    bool stdout_output = false, inject_frb = false, single_pass = false, profile_pass = false, no_control = false;
    bool dump_headers = false;
    int write_fb = 2, nbit = 2, npol = 1, rfi_mode = 2, gpu_id = 0, taps = 1, rows_per_seg = 1024;
    int fft_backend = PB_FFT_LDS, control_port = MC_READER_PORT;
    std::vector<std::string> replay;
    std::string datadir = DEF_DATADIR, logdir = DEF_LOGDIR, out_sink, co_sink;
};

// multilog stand-in: timestamped lines to the per-process log file and optionally stdout
struct Log {
    std::vector<FILE *> fps;
    void open(const std::string &logdir, bool to_stdout)
    {
        char stamp[32], host[256] = "host";
        time_t t = time(nullptr);
        tm g;
        gmtime_r(&t, &g);
        strftime(stamp, sizeof stamp, "%Y%m%d_%H%M%S", &g);
        gethostname(host, sizeof host - 1);
        mkdir(logdir.c_str(), 0777);
        char path[1024];
        snprintf(path, sizeof path, "%s/%s_%s_process_%06d.log", logdir.c_str(), stamp, host, (int)getpid());
        if (FILE *fp = fopen(path, "w")) fps.push_back(fp);
        if (to_stdout) fps.push_back(stdout);
    }
    void line(bool err, const char *fmt, ...) __attribute__((format(printf, 3, 4)))
    {
        char msg[2048], stamp[32];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(msg, sizeof msg, fmt, ap);
        va_end(ap);
        time_t t = time(nullptr);
        tm g;
        gmtime_r(&t, &g);
        strftime(stamp, sizeof stamp, "%Y-%m-%d-%H:%M:%S", &g);
        const size_t n = strlen(msg);
        for (FILE *fp : fps) {
            fprintf(fp, "[%s] %s%s%s", stamp, err ? "ERR: " : "", msg, (n && msg[n - 1] == '\n') ? "" : "\n");
            fflush(fp);
        }
    }
};

// ---- VDIF header (32 bytes, little-endian words; same decoding as the reference's VDIFHeader,
// analysis/baseband.py:19-28) ----
struct VH {
    int64_t second;
    int epoch, frame, threadid, thread, station;
    bool invalid;
};
VH unpack_header(const uint8_t *p)
{
    uint32_t w[4];
    memcpy(w, p, 16);
    VH h;
    h.second = w[0] & 0x3FFFFFFF;
    h.invalid = (w[0] >> 31) != 0;
    h.epoch = (int)((w[1] >> 24) & 0x3F);
    h.frame = (int)(w[1] & 0xFFFFFF);
    h.station = (int)(w[3] & 0xFFFF);
    h.threadid = (int)((w[3] >> 16) & 0x3FF);
    h.thread = h.threadid != 0;
    return h;
}
// Unix time of the start of a VDIF reference epoch (half-years since 2000-01-01)
int64_t epoch_unix(int epoch)
{
    tm g;
    memset(&g, 0, sizeof g);
    g.tm_year = 100 + epoch / 2;
    g.tm_mon = 6 * (epoch % 2);
    g.tm_mday = 1;
    return (int64_t)timegm(&g);
}
int64_t vdif_to_unixepoch(const VH &h) { return epoch_unix(h.epoch) + h.second; }       // src/utils.c:498-514
int64_t frame_mjd(const VH &h) { return 40587 + (epoch_unix(h.epoch) + h.second) / 86400; }
int64_t frame_mjd_sec(const VH &h) { return (epoch_unix(h.epoch) + h.second) % 86400; }
double frame_dmjd(const VH &h, double framepersec)   // getVDIFFrameDMJD (src/process_baseband.cu:241)
{
    return (double)frame_mjd(h) + ((double)frame_mjd_sec(h) + (double)h.frame / framepersec) / 86400.0;
}

// ---- ASCII ring headers: "KEY value" lines; the first token after each key, like ascii_header_get ----
typedef std::vector<std::pair<std::string, std::string>> KV;
std::map<std::string, std::string> ascii_header_parse(const char *raw, size_t n)
{
    std::map<std::string, std::string> out;
    std::string txt(raw, strnlen(raw, n));
    size_t pos = 0;
    while (pos < txt.size()) {
        size_t e = txt.find('\n', pos);
        if (e == std::string::npos) e = txt.size();
        std::string line = txt.substr(pos, e - pos);
        pos = e + 1;
        char k[256], v[1024];
        if (sscanf(line.c_str(), "%255s %1023s", k, v) == 2 && !out.count(k)) out[k] = v;
    }
    return out;
}
std::string ascii_header_format(const KV &kv)
{
    std::string txt;
    char line[2048];
    for (auto &p : kv) {
        snprintf(line, sizeof line, "%-19s %s\n", p.first.c_str(), p.second.c_str());
        txt += line;
    }
    txt.resize(DADA_HDR, '\0');
    return txt;
}
std::string fmt(const char *f, ...) __attribute__((format(printf, 1, 2)));
std::string fmt(const char *f, ...)
{
    char b[1024];
    va_list ap;
    va_start(ap, f);
    vsnprintf(b, sizeof b, f, ap);
    va_end(ap);
    return b;
}
std::string get(const std::map<std::string, std::string> &m, const char *k, const char *dflt = "")
{
    auto it = m.find(k);
    return it == m.end() ? dflt : it->second;
}

// ---- SIGPROC header (src/util.c:51-82 send_string / send_int / send_double; keys and values
// src/process_baseband.cu:226-270) ----
void send_string(std::string &o, const std::string &s)
{
    int32_t n = (int32_t)s.size();
    o.append((const char *)&n, 4);
    o += s;
}
void send_int(std::string &o, const char *name, int32_t v)
{
    send_string(o, name);
    o.append((const char *)&v, 4);
}
void send_double(std::string &o, const char *name, double v)
{
    send_string(o, name);
    o.append((const char *)&v, 8);
}
// all intermediates are C floats, as the reference writes them (:249-259)
float sigproc_angle(double x)
{
    float hh = (float)x;
    float mm = (hh - (int)hh) * 60;
    float ss = (mm - (int)mm) * 60;
    float out = (int)hh * 1e4 + (int)mm * 1e2 + ss;
    return out;
}
std::string sigproc_header(int station, double ra, double dec, const std::string &name, double dmjd, int npol, int nbit)
{
    const double chbw = -64. / PB_NCHAN;
    const double tsamp = (double)PB_NFFT / 128000000 * PB_NSCRUNCH;
    std::string o;
    send_string(o, "HEADER_START");
    send_string(o, "source_name");
    send_string(o, name);
    send_int(o, "barycentric", 0);
    send_int(o, "telescope_id", station);
    send_double(o, "src_raj", (double)sigproc_angle((180 / M_PI) * (24. / 360) * ra));
    send_double(o, "src_dej", (double)sigproc_angle((180 / M_PI) * fabs(dec)));   // sign dropped (:258)
    send_int(o, "data_type", 1);
    send_double(o, "fch1", 384 + (PB_CHANMIN - 0.5) * chbw);
    send_double(o, "foff", chbw);
    send_int(o, "nchans", PB_CHANMAX - PB_CHANMIN + 1);
    send_int(o, "nbits", nbit);
    send_double(o, "tstart", dmjd);
    send_double(o, "tsamp", tsamp);
    send_int(o, "nifs", npol);
    send_string(o, "HEADER_END");
    return o;
}
std::string change_extension(const std::string &name, const char *oldext, const char *newext)
{
    size_t i = name.find(oldext);
    return i == std::string::npos ? name + newext : name.substr(0, i) + newext;
}
// get_fbfile / get_cofbfile (:272-304): CHANMIN < 2411 gives the `_muos` infix, the coadd variant is station 99
void fb_names(int64_t unix_s, int station, const std::string &datadir, std::string &fb, std::string &fb_kur,
              std::string &co, std::string &co_kur)
{
    char stamp[32];
    time_t t = (time_t)unix_s;
    tm g;
    gmtime_r(&t, &g);
    strftime(stamp, sizeof stamp, "%Y%m%d_%H%M%S", &g);
    const char *infix = PB_CHANMIN < 2411 ? "_muos" : "";
    fb = fmt("%s/%s%s_ea%02d.fil", datadir.c_str(), stamp, infix, station);
    co = fmt("%s/%s%s_ea%02d.fil", datadir.c_str(), stamp, infix, 99);
    fb_kur = change_extension(fb, ".fil", "_kur.fil");
    co_kur = change_extension(co, ".fil", "_kur.fil");
}
// write_psrdada_header (:136-201), in its order
KV psrdada_out_header(const std::map<std::string, std::string> &in, int npol, int nbit, const std::string &fb_file,
                      int64_t unix_s, int64_t mjd, int64_t mjd_sec)
{
    const double chbw = -64. / PB_NCHAN;
    const double tsamp = (double)PB_NFFT / 128000000 * PB_NSCRUNCH * 1e6;
    const int nchan = PB_CHANMAX - PB_CHANMIN + 1;
    const int station = atoi(get(in, "STATIONID", "0").c_str());
    char utc[64];
    time_t t = (time_t)unix_s;
    tm g;
    gmtime_r(&t, &g);
    strftime(utc, sizeof utc, "%Y-%m-%d-%H:%M:%S", &g);
    KV h;
    h.emplace_back("STATIONID", fmt("%d", station));
    h.emplace_back("BEAM", fmt("%d", station));
    h.emplace_back("RA", fmt("%f", atof(get(in, "RA", "0").c_str())));
    h.emplace_back("DEC", fmt("%f", atof(get(in, "DEC", "0").c_str())));
    h.emplace_back("NAME", get(in, "NAME"));
    h.emplace_back("SCANSTART", fmt("%f", atof(get(in, "SCANSTART", "0").c_str())));
    h.emplace_back("NCHAN", fmt("%d", nchan));
    h.emplace_back("BANDWIDTH", fmt("%f", nchan * chbw));
    h.emplace_back("CFREQ", fmt("%f", 384. + 0.5 * (PB_CHANMIN + PB_CHANMAX - 1) * chbw));
    h.emplace_back("NPOL", fmt("%d", npol));
    h.emplace_back("NBIT", fmt("%d", nbit));
    h.emplace_back("TSAMP", fmt("%f", tsamp));
    h.emplace_back("UTC_START", utc);
    h.emplace_back("UNIXEPOCH", fmt("%f", (double)unix_s));
    h.emplace_back("VDIF_MJD", fmt("%lld", (long long)mjd));
    h.emplace_back("VDIF_SEC", fmt("%lld", (long long)mjd_sec));
    if (!fb_file.empty()) h.emplace_back("SIGPROC_FILE", fb_file);
    return h;
}

// ---- ring endpoints ----
struct ReadRing {
    virtual ~ReadRing() {}
    virtual int64_t next_header(char *dst) = 0;           // DADA_HDR bytes; 0 = no more observations; < 0 error
    virtual int64_t read(void *buf, size_t n) = 0;        // up to n bytes, fewer only at end of data
    virtual void finish_observation() {}
};
struct WriteRing {
    virtual ~WriteRing() {}
    virtual bool write_header(const std::string &hdr) = 0;
    virtual bool write(const void *buf, size_t n) = 0;
    virtual void end_of_data() = 0;
};

// Replay of dump files: each file = 4096-byte ASCII header + VDIF frame stream.  A single thread copies
// ~9 GB/s out of the page cache (28 ms per second of data); large reads of regular files go over 8 threads.
struct FileRing : ReadRing {
    std::vector<std::string> paths;
    int i = -1, fd = -1;
    bool regular = false;
    explicit FileRing(const std::vector<std::string> &p) : paths(p) {}
    ~FileRing()
    {
        if (fd >= 0) close(fd);
    }
    int64_t next_header(char *dst) override
    {
        if (fd >= 0) close(fd);
        fd = -1;
        if (++i >= (int)paths.size()) return 0;
        fd = open(paths[i].c_str(), O_RDONLY);
        if (fd < 0) return -1;
        struct stat st;
        regular = fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
        return read(dst, DADA_HDR) == (int64_t)DADA_HDR ? (int64_t)DADA_HDR : -1;
    }
    int64_t read(void *buf, size_t n) override
    {
        const size_t CH = 8u << 20;
        if (regular && n >= 2 * CH) {
            const off_t pos = lseek(fd, 0, SEEK_CUR);
            struct stat st;
            fstat(fd, &st);
            const size_t want = std::min<size_t>(n, st.st_size > pos ? (size_t)(st.st_size - pos) : 0);
            if (want >= 2 * CH) {
                const int nth = (int)std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency()));
                std::vector<std::thread> th;
                std::vector<size_t> got(nth, 0), span(nth, 0);
                const size_t per = ((want / nth) + 4095) & ~(size_t)4095;
                for (int t = 0; t < nth; ++t)
                    th.emplace_back([&, t]() {
                        size_t a = std::min(want, (size_t)t * per), b = std::min(want, a + per);
                        span[t] = b - a;
                        while (a < b) {
                            ssize_t r = pread(fd, (char *)buf + a, b - a, pos + (off_t)a);
                            if (r < 0 && errno == EINTR) continue;
                            if (r <= 0) break;
                            a += (size_t)r;
                            got[t] += (size_t)r;
                        }
                    });
                for (int t = 0; t < nth; ++t) th[t].join();
                // only the CONTIGUOUS prefix counts: a piece that came up short (file truncated under us, I/O
                // error) ends the read there, so that the frame stream is never assembled around a hole
                size_t total = 0;
                for (int t = 0; t < nth; ++t) {
                    total += got[t];
                    if (got[t] < span[t]) break;
                }
                lseek(fd, pos + (off_t)total, SEEK_SET);
                return (int64_t)total;
            }
        }
        size_t got = 0;
        while (got < n) {
            ssize_t r = ::read(fd, (char *)buf + got, n - got);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;
            got += (size_t)r;
        }
        return (int64_t)got;
    }
};

struct FileSink : WriteRing {
    FILE *fp;
    explicit FileSink(const std::string &path) : fp(fopen(path.c_str(), "wb")) {}
    ~FileSink()
    {
        if (fp) fclose(fp);
    }
    bool write_header(const std::string &hdr) override { return fp && fwrite(hdr.data(), 1, hdr.size(), fp) == hdr.size(); }
    bool write(const void *buf, size_t n) override { return fp && fwrite(buf, 1, n, fp) == n; }
    void end_of_data() override
    {
        if (fp) fflush(fp);
    }
};

// psrdada rings through the flat C shim of include/pb_dada.h, bound at run time
struct DadaApi {
    void *lib = nullptr;
    decltype(&pb_dada_open) open = nullptr;
    decltype(&pb_dada_next_header) next_header = nullptr;
    decltype(&pb_dada_read) read = nullptr;
    decltype(&pb_dada_read_mt) read_mt = nullptr;
    decltype(&pb_dada_end_read) end_read = nullptr;
    decltype(&pb_dada_write_header) write_header = nullptr;
    decltype(&pb_dada_write) write = nullptr;
    decltype(&pb_dada_end_write) end_write = nullptr;
    decltype(&pb_dada_close) close = nullptr;
    bool load(std::string &why)
    {
        if (lib) return true;
        std::vector<std::string> cand;
        if (const char *e = getenv("PB_DADA_LIB")) cand.push_back(e);
        char self[4096];
        ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
        if (n > 0) {
            self[n] = 0;
            std::string d(self);
            cand.push_back(d.substr(0, d.rfind('/')) + "/libpb_dada.so");
        }
        cand.push_back("libpb_dada.so");
        for (auto &c : cand)
            if ((lib = dlopen(c.c_str(), RTLD_NOW))) break;
        if (!lib) {
            const char *e = dlerror();   // (a second call returns NULL)
            why = "psrdada rings need the shim library (make -C vlite-fast_amd/csrc dada PSRDADA=<prefix>; or "
                  "PB_DADA_LIB=<path to libpb_dada.so>): " + std::string(e ? e : "not found");
            return false;
        }
#define PB_SYM(name) name = (decltype(name))dlsym(lib, "pb_dada_" #name)
        PB_SYM(open); PB_SYM(next_header); PB_SYM(read); PB_SYM(read_mt); PB_SYM(end_read);
        PB_SYM(write_header); PB_SYM(write); PB_SYM(end_write); PB_SYM(close);
#undef PB_SYM
        if (!open || !next_header || !read || !read_mt || !end_read || !write_header || !write || !end_write || !close) {
            why = "the shim library lacks symbols of include/pb_dada.h";
            return false;
        }
        return true;
    }
} g_dada;

struct DadaRing : ReadRing, WriteRing {
    pb_dada *d = nullptr;
    bool header_posted = false;
    static DadaRing *connect(unsigned key, int mode, std::string &why)
    {
        if (!g_dada.load(why)) return nullptr;
        char err[512] = "";
        pb_dada *d = g_dada.open(key, mode, err, sizeof err);
        if (!d) {
            why = err;
            return nullptr;
        }
        DadaRing *r = new DadaRing;
        r->d = d;
        return r;
    }
    ~DadaRing()
    {
        if (d) g_dada.close(d);
    }
    // Data reads: block level by default (the ring's filled buffers copied out by PB_DADA_THREADS threads, 8 unless
    // set; one ipcio_read memcpy thread moves ~9 GB/s = 28 ms per second of data), the reference's ipcio_read for
    // every read with PB_DADA_THREADS=1.  Decided once: psrdada does not let the two alternate on a ring.
    int threads = getenv("PB_DADA_THREADS") ? std::max(1, atoi(getenv("PB_DADA_THREADS"))) : 8;
    int64_t next_header(char *dst) override { return g_dada.next_header(d, dst); }
    int64_t read(void *buf, size_t n) override
    {
        size_t got = 0;
        while (got < n) {
            int64_t r = threads > 1 ? g_dada.read_mt(d, (char *)buf + got, n - got, threads)
                                    : g_dada.read(d, (char *)buf + got, n - got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        return (int64_t)got;
    }
    void finish_observation() override { g_dada.end_read(d); }
    bool write_header(const std::string &hdr) override
    {
        header_posted = true;
        return g_dada.write_header(d, hdr.data()) == 0;
    }
    bool write(const void *buf, size_t n) override { return g_dada.write(d, buf, n) == (int64_t)n; }
    void end_of_data() override
    {
        if (header_posted) g_dada.end_write(d);
        header_posted = false;
    }
};

// ---- control socket: non-blocking membership of 224.3.29.71:20000 (src/utils.c:619, process_baseband.cu:764) ----
// port != MC_READER_PORT (test hook --control-port): a plain UDP socket on 127.0.0.1, no multicast group
int open_control_socket(int port)
{
    int s = socket(AF_INET, SOCK_DGRAM, IPPROTO_UDP);
    if (s < 0) return -1;
    int one = 1;
    setsockopt(s, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in a;
    memset(&a, 0, sizeof a);
    a.sin_family = AF_INET;
    a.sin_port = htons((uint16_t)port);
    a.sin_addr.s_addr = htonl(port == MC_READER_PORT ? INADDR_ANY : INADDR_LOOPBACK);
    ip_mreq m;
    m.imr_multiaddr.s_addr = inet_addr(MC_GROUP);
    m.imr_interface.s_addr = htonl(INADDR_ANY);
    if (bind(s, (sockaddr *)&a, sizeof a) < 0 ||
        (port == MC_READER_PORT && setsockopt(s, IPPROTO_IP, IP_ADD_MEMBERSHIP, &m, sizeof m) < 0)) {
        close(s);
        return -1;
    }
    fcntl(s, F_SETFL, fcntl(s, F_GETFL, 0) | O_NONBLOCK);
    return s;
}
// src/utils.c:174-186: one non-blocking read of up to 32 bytes, any byte equal to cmd
bool test_for_cmd(int sock, char cmd)
{
    if (sock < 0) return false;
    char buf[32];
    ssize_t n = recv(sock, buf, sizeof buf, 0);
    for (ssize_t i = 0; i < n; ++i)
        if (buf[i] == cmd) return true;
    return false;
}

// src/process_baseband.cu:893-913: position, then NAME, then DATAID (check_coords / check_name / check_id)
bool source_allowed(const std::map<std::string, std::string> &hdr)
{
    const std::string name = get(hdr, "NAME"), id = get(hdr, "DATAID");
    g_allow.load();
    const double ra = atof(get(hdr, "RA").c_str()), dec = atof(get(hdr, "DEC").c_str());
    for (const auto &c : g_allow.coords) {
        const double dde = dec - c.second, dra = (ra - c.first) * cos(c.second);      // coord_dist, src/util.c:127-134
        if (sqrt(dde * dde + dra * dra) < 0.01) return true;
    }
    for (const std::string &n : g_allow.names)
        if (name.find(n) != std::string::npos) return true;
    for (const std::string &n : g_allow.ids)
        if (id.find(n) != std::string::npos) return true;
    return false;
}

void usage()
{
    puts("usage: process_baseband -k <key_in hex> -K <key_out hex> -C <key_co hex> -w <0|1|2> -b <2|4|8> -P <1|2>\n"
         "       -r <0|1|2> -g <gpu> [-o] [-i] [-s] [-t] [-m] [-p N]\n"
         "       [--replay FILE...] [--out-sink FILE] [--co-sink FILE] [--datadir DIR] [--logdir DIR] [--no-control]\n"
         "       [--fft-backend lds|hipfft] [--taps 1|4] [--rows-per-seg N] [--dump-headers]\n"
         "       process_baseband --check-source NAME DATAID RA DEC");
}

bool parse(int argc, char **argv, Args &a)
{
    enum { O_REPLAY = 1000, O_DATADIR, O_LOGDIR, O_OUTSINK, O_COSINK, O_NOCTL, O_FFT, O_TAPS, O_ROWS, O_DUMP, O_CTLPORT };
    static option lo[] = {{"replay", required_argument, nullptr, O_REPLAY}, {"datadir", required_argument, nullptr, O_DATADIR},
                          {"logdir", required_argument, nullptr, O_LOGDIR},  {"out-sink", required_argument, nullptr, O_OUTSINK},
                          {"co-sink", required_argument, nullptr, O_COSINK}, {"no-control", no_argument, nullptr, O_NOCTL},
                          {"fft-backend", required_argument, nullptr, O_FFT}, {"taps", required_argument, nullptr, O_TAPS},
                          {"rows-per-seg", required_argument, nullptr, O_ROWS}, {"dump-headers", no_argument, nullptr, O_DUMP},
                          {"control-port", required_argument, nullptr, O_CTLPORT},
                          {nullptr, 0, nullptr, 0}};
    int c;
    while ((c = getopt_long(argc, argv, "hk:K:C:omiw:b:P:r:stg:p:", lo, nullptr)) != -1) {
        switch (c) {
        case 'h': usage(); exit(0);
        case 'k': a.key_in = (unsigned)strtoul(optarg, nullptr, 16); break;     // sscanf "%x" (:367-383)
        case 'K': a.key_out = (unsigned)strtoul(optarg, nullptr, 16); break;
        case 'C': a.key_co = (unsigned)strtoul(optarg, nullptr, 16); break;
        case 'o': a.stdout_output = true; break;
        case 'm': break;                                                         // accepted, no effect (:403-407)
        case 'i': a.inject_frb = true; break;
        case 'w': a.write_fb = atoi(optarg); break;
        case 'b': a.nbit = atoi(optarg); break;
        case 'P': a.npol = atoi(optarg); break;
        case 'r': a.rfi_mode = atoi(optarg); break;
        case 's': a.single_pass = true; break;
        case 't': a.profile_pass = true; break;
        case 'g': a.gpu_id = atoi(optarg); break;
        case 'p': break;                                                         // legacy, ignored
        case O_REPLAY:
            a.replay.push_back(optarg);
            while (optind < argc && argv[optind][0] != '-') a.replay.push_back(argv[optind++]);
            break;
        case O_DATADIR: a.datadir = optarg; break;
        case O_LOGDIR: a.logdir = optarg; break;
        case O_OUTSINK: a.out_sink = optarg; break;
        case O_COSINK: a.co_sink = optarg; break;
        case O_NOCTL: a.no_control = true; break;
        case O_FFT:
            if (!strcmp(optarg, "lds")) a.fft_backend = PB_FFT_LDS;
            else if (!strcmp(optarg, "hipfft")) a.fft_backend = PB_FFT_HIPFFT;
            else return false;
            break;
        case O_TAPS: a.taps = atoi(optarg); break;
        case O_ROWS: a.rows_per_seg = atoi(optarg); break;
        case O_DUMP: a.dump_headers = true; break;
        case O_CTLPORT: a.control_port = atoi(optarg); break;
        default: return false;
        }
    }
    if (a.nbit != 2 && a.nbit != 4 && a.nbit != 8) { fputs("Unsupported NBIT!\n", stderr); return false; }
    if (a.rfi_mode < 0 || a.rfi_mode > 2) { fputs("Unsupported RFI mode!\n", stderr); return false; }
    if (a.npol != 1 && a.npol != 2) { fputs("Unsupported npol!\n", stderr); return false; }
    if (a.rows_per_seg == 1024 && (a.gpu_id < 0 || a.gpu_id > 7)) { fputs("Unsupported GPU id!\n", stderr); return false; }
    return true;
}

void hex(const char *what, const std::string &s)
{
    printf("%s %zu ", what, s.size());
    for (unsigned char ch : s) printf("%02x", ch);
    putchar('\n');
}

int run(const Args &args)
{
    Log log;
    log.open(args.logdir, args.stdout_output);
    const int R = args.rows_per_seg;
    const size_t frames_per_sec = (size_t)R * SEG_PER_SEC * 12500 / VD_DAT;   // per thread; 25600 at R = 1024
    const size_t sec_bytes = 2 * frames_per_sec * VD_FRM;                      // 257 638 400 at R = 1024
    std::string why;
    ReadRing *in_ring = nullptr;
    if (!args.replay.empty()) in_ring = new FileRing(args.replay);
    else if (!(in_ring = DadaRing::connect(args.key_in, PB_DADA_READ, why))) {
        log.line(true, "could not connect to input ring %x: %s", args.key_in, why.c_str());
        return 1;
    }
    WriteRing *out_ring = nullptr, *co_ring = nullptr;
    if (!args.dump_headers) {
        if (!args.out_sink.empty()) out_ring = new FileSink(args.out_sink);
        else if (args.key_out && !(out_ring = DadaRing::connect(args.key_out, PB_DADA_WRITE, why))) {
            log.line(true, "could not connect to output ring %x: %s", args.key_out, why.c_str());
            return 1;
        }
        if (!args.co_sink.empty()) co_ring = new FileSink(args.co_sink);
        else if (args.key_co && !(co_ring = DadaRing::connect(args.key_co, PB_DADA_WRITE, why))) {
            log.line(true, "could not connect to coadd ring %x: %s", args.key_co, why.c_str());
            return 1;
        }
    }
    pb_handle *h = nullptr;
    pb_sizes sz;
    memset(&sz, 0, sizeof sz);
    const int nsets = 2;
    if (!args.dump_headers) {
        pb_config cfg;
        pb_config_default(&cfg);
        cfg.device = args.gpu_id;
        cfg.nant = 1;
        cfg.nbit = args.nbit;
        cfg.npol = args.npol;
        cfg.rfi_mode = args.rfi_mode;
        cfg.taps = args.taps;
        cfg.fft_backend = args.fft_backend;
        cfg.rows_per_seg = R;
        cfg.max_seg = SEG_PER_SEC;
        cfg.inject_frb = args.inject_frb;
        cfg.nsets = nsets;
        if (pb_create(&cfg, &h) != PB_OK) {
            log.line(true, "pb_create: %s", pb_last_error(nullptr));
            return 1;
        }
        pb_query(h, &sz);
    }
    const size_t trim = sz.code_bytes_per_seg;
    const int ctl = args.no_control ? -1 : open_control_socket(args.control_port);
    std::vector<uint8_t *> blocks;        // page-locked staging for one second of frames
    // One second of frames is assembled in page-locked memory (nsets + 1 buffers in turn: the H2D of second k is
    // known to be complete once its filterbank bytes have been fetched, which happens before buffer
    // k mod (nsets + 1) comes round again), read straight into place -- one copy from the ring.  Allocated at
    // start-up, as the reference allocates its host buffer (cudaMallocHost, :579): pinning 0.77 GB takes 0.15 s.
    if (h)
        for (int i = 0; i < nsets + 1; ++i) {
            uint8_t *b = (uint8_t *)pb_host_alloc(sec_bytes);
            if (!b) {
                log.line(true, "pb_host_alloc(%zu) failed", sec_bytes);
                return 1;
            }
            blocks.push_back(b);
        }
    const double tsamp = 12500.0 / 128e6 * 8;
    int exit_status = 0;
    bool quit = false;
#define PBCHK(call)                                                            \
    do {                                                                       \
        if ((call) != PB_OK) {                                                 \
            log.line(true, "%s: %s", #call, pb_last_error(h));                 \
            return 1;                                                          \
        }                                                                      \
    } while (0)

    while (!quit) {
        if (test_for_cmd(ctl, CMD_QUIT)) break;
        log.line(false, "Waiting for DADA header.");
        std::vector<char> raw_hdr(DADA_HDR);
        const int64_t hn = in_ring->next_header(raw_hdr.data());
        if (hn <= 0) {
            log.line(false, "Input ring closed.  Exiting.");
            break;
        }
        const double t_obs = now();
        auto hdr = ascii_header_parse(raw_hdr.data(), DADA_HDR);
        log.line(false, "Beginning new observation.");
        if (h) PBCHK(pb_reset_history(h, 0));           // taps=4: the FIR window does not span observations
        uint8_t first[VD_FRM];
        if (in_ring->read(first, VD_FRM) != (int64_t)VD_FRM) {
            log.line(true, "Problem reading first bloody frame!  Bailing.");
            return 1;
        }
        VH vh = unpack_header(first);
        if (vh.frame != 0 && vh.thread != 0) {          // sic: `&&` in the reference (:845)
            log.line(true, "Incoming data were not aligned!");
            return 1;
        }
        const int station = atoi(get(hdr, "STATIONID", "0").c_str());
        const int64_t t_unix = vdif_to_unixepoch(vh);
        std::string fb, fb_kur, cofb, cofb_kur;
        fb_names(t_unix, station, args.datadir, fb, fb_kur, cofb, cofb_kur);
        const std::string heimdall_file = args.rfi_mode ? fb_kur : fb, coheimdall_file = args.rfi_mode ? cofb_kur : cofb;
        const int64_t mjd = frame_mjd(vh), mjd_sec = frame_mjd_sec(vh);
        const std::string sp_hdr = sigproc_header(station, atof(get(hdr, "RA", "0").c_str()), atof(get(hdr, "DEC", "0").c_str()),
                                                  get(hdr, "NAME"), frame_dmjd(vh, (double)frames_per_sec), args.npol, args.nbit);
        const std::string out_hdr = ascii_header_format(psrdada_out_header(hdr, args.npol, args.nbit, heimdall_file, t_unix, mjd, mjd_sec));
        const std::string co_hdr = ascii_header_format(psrdada_out_header(hdr, args.npol, args.nbit, coheimdall_file, t_unix, mjd, mjd_sec));
        if (args.dump_headers) {
            hex("sigproc", sp_hdr);
            hex("out", out_hdr);
            hex("co", co_hdr);
            printf("files %s %s %s %s\n", fb.c_str(), fb_kur.c_str(), cofb.c_str(), cofb_kur.c_str());
            return 0;
        }
        bool write_to_null = args.write_fb == 0;
        if (args.write_fb == 1) {
            g_allow.load();
            if (g_allow.ok)
                log.line(false, "Source list for -w 1: %s (%zu names, %zu dataids, %zu positions).", g_allow.path.c_str(),
                         g_allow.names.size(), g_allow.ids.size(), g_allow.coords.size());
            else
                log.line(true, "Source list for -w 1 not usable (%s): no source will be recorded.", g_allow.path.c_str());
            if (source_allowed(hdr))
                log.line(false, "Source %s matches target list, recording filterbank data.", get(hdr, "NAME").c_str());
            else {
                write_to_null = true;
                log.line(false, "Source %s not on target list, disabling filterbank data.", get(hdr, "NAME").c_str());
            }
        }
        if (write_to_null) {
            log.line(false, "Filterbank output disabled.  Would have written to %s.", fb.c_str());
            fb = fb_kur = "/dev/null";
        }
        FILE *fb_fp = nullptr, *fb_kur_fp = nullptr;
        if (args.rfi_mode == 0 || args.rfi_mode == 2) {
            fb_fp = fopen(fb.c_str(), "wb");
            log.line(false, "Writing no-RFI-excision filterbanks to %s.", fb.c_str());
        } else {
            fb_fp = fopen(fb_kur.c_str(), "wb");
            log.line(false, "Writing RFI-excision filterbanks to %s.", fb_kur.c_str());
        }
        if (args.rfi_mode == 2) {
            fb_kur_fp = fopen(fb_kur.c_str(), "wb");
            log.line(false, "Writing RFI-excision filterbanks to %s.", fb_kur.c_str());
        }
        if (!fb_fp || (args.rfi_mode == 2 && !fb_kur_fp)) {
            log.line(true, "could not open the filterbank files in %s", args.datadir.c_str());
            return 1;
        }
        // a failed ring write is fail-stop, like check_ipcio_write (src/process_baseband.cu:322-332)
        auto ring_write = [&](WriteRing *r, const void *buf, size_t n) -> int {
            if (r->write(buf, n)) return 0;
            fprintf(stderr, "failed ipcio write\n");
            log.line(true, "Tried to write %lu bytes to psrdada buffer but the write came up short.", (unsigned long)n);
            return 1;
        };
        if ((out_ring && !out_ring->write_header(out_hdr)) || (co_ring && !co_ring->write_header(co_hdr))) {
            log.line(true, "Could not write the psrdada output header.");
            return 1;
        }
        fwrite(sp_hdr.data(), 1, sp_hdr.size(), fb_fp);
        if (fb_kur_fp) fwrite(sp_hdr.data(), 1, sp_hdr.size(), fb_kur_fp);

        int64_t current_sec = vh.second;
        log.line(false, "Starting sec=%lld, thread=%d", (long long)current_sec, vh.thread);
        long integrated_sec = 0, queued = 0;
        size_t fb_bytes = 0;
        double t_rt = now(), prof_read = 0, prof_todev = 0, prof_write = 0;
        std::vector<std::vector<uint8_t>> out_buf;      // 10-s buffer of the stream heimdall gets (:691-697)
        if (args.profile_pass) {
            pb_timers tm;
            pb_get_timers(h, &tm, 1);
            pb_profile(h, 1);
        }
        // filterbank bytes of queued second k -> files and rings (:1364-1441, :1482-1494)
        auto collect = [&](long k) -> int {
            const double t0 = now();
            PBCHK(pb_select_set(h, (int)(k % nsets)));
            const uint8_t *raw = nullptr, *kur = nullptr;
            if (args.rfi_mode != 1) PBCHK(pb_fetch_ptr(h, 0, 0, &raw));
            if (args.rfi_mode != 0) PBCHK(pb_fetch_ptr(h, 0, 1, &kur));
            const uint8_t *main_codes = args.rfi_mode != 1 ? raw : kur, *heim_codes = args.rfi_mode != 0 ? kur : raw;
            for (int iseg = 0; iseg < SEG_PER_SEC; ++iseg) {
                if (co_ring && ring_write(co_ring, heim_codes + iseg * trim, trim)) return 1;
                fwrite(main_codes + iseg * trim, 1, trim, fb_fp);
                if (fb_kur_fp) fwrite(kur + iseg * trim, 1, trim, fb_kur_fp);
                fb_bytes += trim;
            }
            out_buf.emplace_back(heim_codes, heim_codes + SEG_PER_SEC * trim);
            ++integrated_sec;
            if (integrated_sec % 10 == 0) {
                const double lag = (now() - t_rt) - 10.0 * (R / 1024.0);
                if (lag > 0.5) log.line(true, "Measured time exceeding integrated time: lag %.2f s", lag);
                t_rt = now();
            }
            if (integrated_sec >= 10 && out_ring) {
                if (integrated_sec == 10) {                                        // the full 10-s buffer, one write
                    std::vector<uint8_t> all;
                    for (auto &b : out_buf) all.insert(all.end(), b.begin(), b.end());
                    if (ring_write(out_ring, all.data(), all.size())) return 1;
                } else if (ring_write(out_ring, out_buf.back().data(), out_buf.back().size()))  // then 1 s at a time
                    return 1;
            }
            if (out_buf.size() > 10) out_buf.erase(out_buf.begin());
            prof_write += now() - t0;
            return 0;
        };

        // Frames are placed by their own headers (thread id, frame number: :1017-1034), a second is closed by the
        // first frame of another second (:1019, :1058) and dispatched only then, and a frame that never arrives
        // leaves zeros.  The frames of one second are gathered in a page-locked block in bulk; when frames were
        // dropped the block swallows the head of the next second, which is carried over, so that the stream
        // re-aligns at once.  Seconds are pipelined over the handle's buffer sets: second k is queued (H2D +
        // kernels) before the output of second k-1 is collected.
        std::vector<uint8_t> carry(first, first + VD_FRM);
        while (true) {
            double t0 = now();
            uint8_t *block = blocks[queued % blocks.size()];
            size_t have = carry.size();
            memcpy(block, carry.data(), have);
            carry.clear();
            std::vector<uint8_t> boundary;   // bytes (>= one frame) of the next second, already read
            bool eod = false;
            while (boundary.empty() && !eod) {
                if (have < sec_bytes) {
                    int64_t got = in_ring->read(block + have, sec_bytes - have);
                    if (got < 0) got = 0;
                    if (got % VD_FRM) {
                        log.line(false, "Packet size=%d, expected %d.  Aborting this observation.", (int)(got % VD_FRM), (int)VD_FRM);
                        got -= got % VD_FRM;
                        eod = true;
                    } else if ((size_t)got < sec_bytes - have) {
                        eod = true;
                    }
                    have += (size_t)got;
                } else {
                    uint8_t nxt[VD_FRM];   // block full and all of this second: the next frame decides
                    const int64_t n = in_ring->read(nxt, VD_FRM);
                    if (n != (int64_t)VD_FRM) {
                        if (n > 0) log.line(false, "Packet size=%d, expected %d.  Aborting this observation.", (int)n, (int)VD_FRM);
                        eod = true;
                        break;
                    }
                    if (unpack_header(nxt).second != current_sec) boundary.assign(nxt, nxt + VD_FRM);
                    continue;              // (a duplicate frame of this second: dropped)
                }
                const size_t nfr = have / VD_FRM;
                size_t cut = nfr;
                for (size_t f = 0; f < nfr; ++f) {
                    uint32_t w0;
                    memcpy(&w0, block + f * VD_FRM, 4);
                    if ((int64_t)(w0 & 0x3FFFFFFF) != current_sec) {
                        cut = f;
                        break;
                    }
                }
                if (cut < nfr) {
                    boundary.assign(block + cut * VD_FRM, block + have);
                    // what follows the cut is not this second's: hide it from the device-side frame index
                    for (size_t f = cut; f < nfr; ++f) block[f * VD_FRM + 3] |= 0x80;   // VDIF invalid-data bit
                    have = cut * VD_FRM;
                }
            }
            prof_read += now() - t0;
            if (boundary.empty()) break;   // end of data: the last (partial or whole) second is dropped
            const VH nh = unpack_header(boundary.data());
            if (nh.second - current_sec > 1) {
                log.line(true, "Major data skip!  (%lld vs. %lld; thread = %d) Aborting this observation.",
                         (long long)nh.second, (long long)current_sec, nh.thread);
                exit_status = 1;
                quit = true;
                break;
            }
            if (test_for_cmd(ctl, CMD_QUIT)) {
                log.line(false, "Received CMD_QUIT, indicating data taking is ceasing.  Exiting.");
                quit = true;
                break;
            }
            // frames were dropped: whatever an earlier second left in the tail of the block is not data
            for (size_t o = have; o + VD_FRM <= sec_bytes; o += VD_FRM) block[o + 3] |= 0x80;
            const int inject_now = (args.inject_frb && current_sec % 60 == 0) ? 1 : 0;
            if (inject_now) log.line(false, "Injecting an FRB with integrated = %.2f!!!.", (double)queued);
            t0 = now();
            PBCHK(pb_select_set(h, (int)(queued % nsets)));
            PBCHK(pb_submit_vdif_at(h, 0, 0, block, sec_bytes, current_sec, 0));
            PBCHK(pb_process(h, SEG_PER_SEC, inject_now));
            prof_todev += now() - t0;
            ++queued;
            if (queued - integrated_sec >= nsets)
                if (collect(integrated_sec)) return 1;   // the oldest queued second, while the newest computes
            current_sec = nh.second;
            carry.swap(boundary);
        }
        while (integrated_sec < queued)
            if (collect(integrated_sec)) return 1;

        if (out_ring) out_ring->end_of_data();
        if (co_ring) co_ring->end_of_data();
        in_ring->finish_observation();
        fclose(fb_fp);
        if (fb_kur_fp) fclose(fb_kur_fp);
        const size_t nsamp = fb_bytes * (8 / args.nbit) / 4096;
        log.line(false, "Wrote %.2f MB (%.2f s) to %s", fb_bytes * 1e-6, nsamp * tsamp, fb.c_str());
        log.line(false, "Proc Time...%.3f", now() - t_obs);
        if (args.profile_pass) {
            // the reference's PROFILE block (:1538-1556).  Its kernels are fused here: "Convert" lives in the
            // kurtosis and channeliser kernels, "FFT" is the channeliser (unpack + FFT + detect), and normalise /
            // pscrunch / tscrunch / digitise are one kernel, reported under "Normalize".
            pb_profile(h, 0);
            pb_timers tm;
            pb_get_timers(h, &tm, 1);
            log.line(false, "Read Time...%.3f", prof_read);
            log.line(false, "Copy To Dev.%.3f", prof_todev);
            log.line(false, "Kurtosis....%.3f", tm.ms[PB_ST_KURTOSIS] * 1e-3);
            log.line(false, "FFT.........%.3f", (tm.ms[PB_ST_CHANNELIZE] + tm.ms[PB_ST_FFT] + tm.ms[PB_ST_INJECT]) * 1e-3);
            log.line(false, "Normalize...%.3f", tm.ms[PB_ST_DETECT] * 1e-3);
            log.line(false, "Write.......%.3f", prof_write);
        }
        if (args.profile_pass || args.single_pass) break;
    }
    if (h) {
        pb_sync(h);
        for (uint8_t *b : blocks) pb_host_free(b);
        pb_destroy(h);
    }
    delete in_ring;
    delete out_ring;
    delete co_ring;
    return exit_status;
}

}   // namespace

int main(int argc, char **argv)
{
    // --check-source NAME DATAID RA DEC: the -w 1 decision for one header (prints 1 or 0; no GPU, no rings) -- what
    // the reference's check_coords / check_name / check_id answer (tests/test_ref_util.py compares)
    if (argc == 6 && !strcmp(argv[1], "--check-source")) {
        std::map<std::string, std::string> hdr{{"NAME", argv[2]}, {"DATAID", argv[3]}, {"RA", argv[4]}, {"DEC", argv[5]}};
        printf("%d\n", source_allowed(hdr) ? 1 : 0);
        return 0;
    }
    Args args;
    if (!parse(argc, argv, args)) {
        usage();
        return 1;
    }
    return run(args);
}
