// Streaming FASTQ reader with the record semantics the reference gets from
// btllib::SeqReader (LONG_MODE, default flags): id = header up to the first
// whitespace, sequence case-folded to upper case, 4-line records, records in
// file order.  Plain or gzip-compressed text (zlib; btllib::SeqReader pipes compressed
// input through an external decompressor).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace gr {

// the input file: a plain regular file (pread, large requests by several threads) or anything
// zlib reads — gzip data, pipes (GRP_ZLIB_READER=1 sends plain files through zlib too)
// An input that could not be read to its end (EIO, a file that shrank under the run, a damaged gzip stream) must not
// pass for a shorter input: the readers below note it here (first message wins) and the program ends with an
// error behind the pass that met it instead of classifying a prefix of the reads silently.
void note_input_failure(const std::string& what);
bool input_failed(std::string* what = nullptr);
void clear_input_failure();

class InputFile
{
public:
  explicit InputFile(const std::string& path);
  ~InputFile();
  InputFile(const InputFile&) = delete;
  InputFile& operator=(const InputFile&) = delete;
  bool ok() const { return opened_; } // the file could be opened (it stays true behind the end of the data)
  size_t read(char* dst, size_t n); // 0 at the end of the data (or on a read error)
  int peek();                      // next byte without consuming it, -1 at the end
  uint64_t plain_size() const { return size_; } // bytes of a plain regular file, 0: unknown (gzip data, a pipe)

private:
  void* f_ = nullptr; // gzFile; closed (nullptr) once the data has ended, so that a truncated stream is noticed
  bool opened_ = false;
  int fd_ = -1;       // plain regular file
  uint64_t off_ = 0;  // ... and the read position in it
  uint64_t size_ = 0;
  std::string path_;
};

struct RecordRef
{
  size_t id_off, id_len;
  size_t seq_off, seq_len;
  size_t qual_off, qual_len;
};

// a run of consecutive records; all text lives in one buffer
struct RecordBatch
{
  std::vector<char> text;
  std::vector<RecordRef> rec;
  size_t bases = 0;
  void clear()
  {
    text.clear();
    rec.clear();
    bases = 0;
  }
  const char* id(size_t i) const { return text.data() + rec[i].id_off; }
  const char* seq(size_t i) const { return text.data() + rec[i].seq_off; }
  const char* qual(size_t i) const { return text.data() + rec[i].qual_off; }
  std::string id_str(size_t i) const { return std::string(id(i), rec[i].id_len); }
};

class FastqStream
{
public:
  explicit FastqStream(const std::string& path);
  ~FastqStream();
  bool ok() const { return in_.ok(); }
  // SeqReader::get_format() == FASTQ  <=>  first byte of the file is '@'
  bool is_fastq();
  // appends up to max_records / ~max_bases to `out` (cleared first); false at EOF
  bool next_batch(RecordBatch& out, size_t max_records, size_t max_bases);

private:
  bool fill();
  bool get_line(const char*& p, size_t& n); // without the trailing newline / CR / blanks
  InputFile in_;
  std::vector<char> buf_;
  size_t pos_ = 0, end_ = 0;
  bool eof_ = false;
  bool stopped_ = false; // a line that is not a FASTQ header was met: end of input for good (as the GPU ingest)
  std::vector<char> carry_;
};

} // namespace gr
