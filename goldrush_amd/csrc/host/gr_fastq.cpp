#include "gr_fastq.hpp"

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/stat.h>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <zlib.h>

namespace gr {

// A plain regular file is read with pread: large requests are cut into slices that several
// threads copy out of the page cache at once (one thread moves ~6 GB/s, the GPU ingest
// behind it parses > 20 GB/s).  Everything else — gzip data, pipes — goes through zlib.
InputFile::InputFile(const std::string& path)
{
  path_ = path;
  const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
  if (fd >= 0) {
    struct stat st;
    unsigned char magic[2] = { 0, 0 };
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && !getenv("GRP_ZLIB_READER")) {
      const ssize_t k = pread(fd, magic, 2, 0);
      if (k >= 0 && !(k == 2 && magic[0] == 0x1f && magic[1] == 0x8b)) {
        fd_ = fd;
        size_ = (uint64_t)st.st_size;
        opened_ = true;
        return;
      }
    }
    ::close(fd);
  }
  gzFile f = gzopen(path.c_str(), "rb");
  if (f) {
    gzbuffer(f, 1u << 20);
  }
  f_ = f;
  opened_ = f != nullptr;
}

InputFile::~InputFile()
{
  if (f_) {
    gzclose(static_cast<gzFile>(f_));
  }
  if (fd_ >= 0) {
    ::close(fd_);
  }
}

namespace {
std::mutex g_fail_mu;
std::string g_fail_what;
bool g_failed = false;
} // namespace

void
note_input_failure(const std::string& what)
{
  std::lock_guard<std::mutex> g(g_fail_mu);
  if (!g_failed) {
    g_failed = true;
    g_fail_what = what;
  }
}

bool
input_failed(std::string* what)
{
  std::lock_guard<std::mutex> g(g_fail_mu);
  if (g_failed && what) {
    *what = g_fail_what;
  }
  return g_failed;
}

void
clear_input_failure()
{
  std::lock_guard<std::mutex> g(g_fail_mu);
  g_failed = false;
  g_fail_what.clear();
}

// *err: errno of a read that failed (0: none; a short count is then the end of the file)
static size_t
pread_all(int fd, char* dst, size_t n, uint64_t off, int* err)
{
  size_t got = 0;
  while (got < n) {
    const ssize_t r = pread(fd, dst + got, n - got, (off_t)(off + got));
    if (r < 0 && errno == EINTR) {
      continue;
    }
    if (r < 0) {
      *err = errno;
      break;
    }
    if (r == 0) {
      break;
    }
    got += (size_t)r;
  }
  return got;
}

size_t
InputFile::read(char* dst, size_t n)
{
  if (fd_ >= 0) {
    constexpr size_t kSlice = size_t(16) << 20;
    static const size_t max_threads = [] { // GRP_READ_THREADS: threads copying one request out of the page cache (default 8)
      const char* e = getenv("GRP_READ_THREADS");
      const long v = e ? atol(e) : 0;
      return (size_t)(v > 0 ? std::min(v, 64l) : 8l);
    }();
    size_t threads = std::min<size_t>(n / kSlice, std::min<size_t>(max_threads, std::max(1u, std::thread::hardware_concurrency())));
    size_t got = 0;
    int err = 0;
    if (threads < 2) {
      got = pread_all(fd_, dst, n, off_, &err);
    } else {
      // slice i covers [i * per, (i + 1) * per); the data ends in the first short slice
      const size_t per = (n + threads - 1) / threads;
      std::vector<size_t> part(threads, 0);
      std::vector<int> errs(threads, 0);
      std::vector<std::thread> pool;
      for (size_t i = 1; i < threads; ++i) {
        pool.emplace_back([&, i] { part[i] = pread_all(fd_, dst + i * per, std::min(per, n - i * per), off_ + i * per, &errs[i]); });
      }
      part[0] = pread_all(fd_, dst, per, off_, &errs[0]);
      for (auto& t : pool) {
        t.join();
      }
      for (size_t i = 0; i < threads; ++i) {
        got += part[i];
        if (part[i] < std::min(per, n - i * per)) {
          err = errs[i];
          break;
        }
      }
    }
    // the size is known: a read that failed, or data that ends in front of it, is an error and not the end of the input
    const uint64_t want = std::min<uint64_t>(n, size_ > off_ ? size_ - off_ : 0);
    if (err != 0 || got < want) {
      note_input_failure("reading " + path_ + " failed at byte " + std::to_string(off_ + got) + " of " + std::to_string(size_) + ": " + (err ? strerror(err) : "the file ends early"));
    }
    off_ += got;
    return got;
  }
  size_t got = 0;
  while (f_ && got < n) { // gzread takes an unsigned length
    const unsigned want = (unsigned)std::min<size_t>(n - got, 1u << 30);
    const int r = gzread(static_cast<gzFile>(f_), dst + got, want);
    if (r < 0) {
      int zerr = 0;
      const char* msg = gzerror(static_cast<gzFile>(f_), &zerr);
      note_input_failure("reading " + path_ + " failed: " + (msg ? msg : "zlib error"));
      break;
    }
    if (r == 0) {
      // the end of the data — or of a gzip member that was cut off: zlib hands out what it could inflate and then
      // reports 0 bytes like at a proper end.  gzerror says Z_BUF_ERROR when the cut was met inside this
      // call, gzclose_r when an earlier call swallowed it; a stream that did not end on a complete member is an error.
      int zerr = 0;
      const char* msg = gzerror(static_cast<gzFile>(f_), &zerr);
      const std::string why = (zerr != Z_OK && msg && *msg) ? msg : "";
      const int rc = gzclose_r(static_cast<gzFile>(f_));
      f_ = nullptr;
      if (zerr != Z_OK || rc != Z_OK) {
        note_input_failure("reading " + path_ + " failed: " + (why.empty() ? "the compressed stream ends early (truncated gzip data)" : why));
      }
      break;
    }
    got += (size_t)r;
  }
  return got;
}

int
InputFile::peek()
{
  if (fd_ >= 0) {
    unsigned char c;
    return pread(fd_, &c, 1, (off_t)off_) == 1 ? (int)c : -1;
  }
  if (!f_) {
    return -1;
  }
  const int c = gzgetc(static_cast<gzFile>(f_));
  if (c >= 0) {
    gzungetc(c, static_cast<gzFile>(f_));
  }
  return c;
}

FastqStream::FastqStream(const std::string& path)
  : in_(path)
{
  buf_.resize(size_t(8) << 20);
}

FastqStream::~FastqStream() {}

bool
FastqStream::fill()
{
  if (eof_ || !in_.ok()) {
    return false;
  }
  // keep the unread tail, refill the rest
  const size_t tail = end_ - pos_;
  if (tail && pos_) {
    memmove(buf_.data(), buf_.data() + pos_, tail);
  }
  pos_ = 0;
  end_ = tail;
  if (end_ == buf_.size()) {
    buf_.resize(buf_.size() * 2); // a single line longer than the buffer
  }
  const size_t got = in_.read(buf_.data() + end_, buf_.size() - end_);
  if (got == 0) {
    eof_ = true;
    return false;
  }
  end_ += got;
  return true;
}

bool
FastqStream::is_fastq()
{
  if (pos_ == end_ && !fill()) {
    return false;
  }
  return buf_[pos_] == '@';
}

bool
FastqStream::get_line(const char*& p, size_t& n)
{
  for (;;) {
    const char* start = buf_.data() + pos_;
    const char* nl = (const char*)memchr(start, '\n', end_ - pos_);
    if (nl) {
      p = start;
      n = (size_t)(nl - start);
      pos_ += n + 1;
      break;
    }
    if (!fill()) {
      if (pos_ == end_) {
        return false;
      }
      p = buf_.data() + pos_; // last line without newline
      n = end_ - pos_;
      pos_ = end_;
      break;
    }
  }
  while (n > 0 && (p[n - 1] == '\r' || p[n - 1] == ' ' || p[n - 1] == '\t')) {
    --n;
  }
  return true;
}

bool
FastqStream::next_batch(RecordBatch& out, size_t max_records, size_t max_bases)
{
  out.clear();
  while (!stopped_ && out.rec.size() < max_records && out.bases < max_bases) {
    const char* p;
    size_t n;
    // get_line pointers die at the next refill, so each line is copied at once
    if (!get_line(p, n)) {
      break;
    }
    if (n == 0 || p[0] != '@') {
      stopped_ = true; // not a FASTQ header: stop like a reader at end of input, for good
      break;           // (grp_fastq_parse latches the same state: both sources see the same records)
    }
    RecordRef r{};
    size_t idn = 0;
    while (1 + idn < n && !isspace((unsigned char)p[1 + idn])) {
      ++idn;
    }
    r.id_off = out.text.size();
    r.id_len = idn;
    out.text.insert(out.text.end(), p + 1, p + 1 + idn);
    out.text.push_back('\0');
    if (!get_line(p, n)) {
      out.text.resize(r.id_off);
      break;
    }
    r.seq_off = out.text.size();
    r.seq_len = n;
    out.text.insert(out.text.end(), p, p + n);
    out.text.push_back('\0');
    for (size_t i = 0; i < n; ++i) {
      char& c = out.text[r.seq_off + i];
      if (c >= 'a' && c <= 'z') {
        c = (char)(c - 32);
      }
    }
    if (!get_line(p, n) || !get_line(p, n)) { // '+' line, then qualities
      out.text.resize(r.id_off);
      break;
    }
    r.qual_off = out.text.size();
    r.qual_len = n;
    out.text.insert(out.text.end(), p, p + n);
    out.text.push_back('\0');
    out.rec.push_back(r);
    out.bases += r.seq_len;
  }
  return !out.rec.empty();
}

} // namespace gr
