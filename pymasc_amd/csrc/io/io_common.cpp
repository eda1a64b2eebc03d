#include "io_common.h"
#include "../../../include/pymasc_amd_io.h"

#include <cerrno>
#include <cstring>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace pmx_io {

static thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

int pick_threads(int requested)
{
    if (requested > 0) return requested > 64 ? 64 : requested;
    unsigned hc = std::thread::hardware_concurrency();
    if (hc == 0) hc = 1;
    return (int)(hc > 16 ? 16 : hc);
}

void MappedFile::open(const char *path)
{
    close();
    const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) throw Error(PMX_IO_ERR_OPEN, std::string("cannot open: ") + strerror(errno));
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        ::close(fd);
        throw Error(PMX_IO_ERR_OPEN, "not a regular file");
    }
    size = (size_t)st.st_size;
    if (size == 0) {
        ::close(fd);
        throw Error(PMX_IO_ERR_FORMAT, "empty file");
    }
    void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) {
        size = 0;
        throw Error(PMX_IO_ERR_OPEN, std::string("mmap failed: ") + strerror(errno));
    }
    madvise(p, size, MADV_SEQUENTIAL);
    data = (const uint8_t *)p;
}

void MappedFile::close()
{
    if (data) munmap((void *)data, size);
    data = nullptr;
    size = 0;
}

}  // namespace pmx_io

extern "C" {
const char *pmx_io_last_error(void) { return pmx_io::g_last_error.c_str(); }
int pmx_io_version(void) { return 1; }
}
