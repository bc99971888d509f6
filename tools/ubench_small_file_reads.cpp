// ubench_small_file_reads.cpp -- what bounds reading MANY SMALL files from several threads of one process (DESIGN.md 4.5:
// "60-80 ms per 100 000 files whether 8 or 16 threads read")?  A tree of n_dirs x per_dir files of 4 KiB in /dev/shm, read
// by T threads, one directory at a time per thread:
//   mode 0  open(path) + pread + close                      (what the reader threads of mi_stage.hip do)
//   mode 1  readdir + fstatat + openat(dirfd) + pread + close   (the walk's lstat, then the read where the file is listed)
//   mode 2  readdir + openat(dirfd, O_NOFOLLOW) + fstat(fd) + pread + close   (one path lookup per file instead of two)
//   mode 3  mode 2 in threads that left the process's file-descriptor table: unshare(CLONE_FILES) + close_range(3, ~0) --
//           open and close then take the THREAD's own table lock, not the one all threads of the process share
//   g++ -O2 -o tools/bin/ubench_small_file_reads tools/ubench_small_file_reads.cpp -lpthread
//   tools/bin/ubench_small_file_reads [n_dirs=200] [per_dir=500]
#include <dirent.h>
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int nd = argc > 1 ? atoi(argv[1]) : 200, per = argc > 2 ? atoi(argv[2]) : 500;
    char tmpl[] = "/dev/shm/mi_ubench_XXXXXX";
    const std::string root = mkdtemp(tmpl);
    std::vector<std::string> dirs;
    std::vector<char> blob(4096, 'x');
    for (int d = 0; d < nd; ++d) {
        char b[64];
        snprintf(b, sizeof b, "/d%04d", d);
        dirs.push_back(root + b);
        mkdir(dirs.back().c_str(), 0755);
        for (int f = 0; f < per; ++f) {
            snprintf(b, sizeof b, "/f%05d", f);
            const int fd = open((dirs.back() + b).c_str(), O_CREAT | O_WRONLY, 0644);
            if (write(fd, blob.data(), blob.size()) != (ssize_t)blob.size()) return 2;
            close(fd);
        }
    }
    for (int mode = 0; mode < 4; ++mode)
        for (int nt : {1, 4, 8, 16, 32}) {
            std::atomic<int> next{0}, unshared{0};
            std::atomic<long> bytes{0};
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&] {
                    if (mode == 3 && unshare(CLONE_FILES) == 0) {
                        syscall(SYS_close_range, 3u, ~0u, 0u);                 // the private table starts empty
                        ++unshared;
                    }
                    std::vector<char> buf(1 << 20);
                    for (int d; (d = next.fetch_add(1)) < nd;) {
                        if (mode == 0) {
                            for (int f = 0; f < per; ++f) {
                                char b[64];
                                snprintf(b, sizeof b, "/f%05d", f);
                                const int fd = open((dirs[d] + b).c_str(), O_RDONLY | O_CLOEXEC);
                                bytes += pread(fd, buf.data(), 4096, 0);
                                close(fd);
                            }
                            continue;
                        }
                        const int dfd = open(dirs[d].c_str(), O_RDONLY | O_DIRECTORY | O_CLOEXEC);
                        DIR* dir = fdopendir(dfd);
                        std::vector<std::string> names;
                        while (dirent* de = readdir(dir))
                            if (de->d_name[0] != '.') names.push_back(de->d_name);
                        std::sort(names.begin(), names.end());
                        for (auto& n : names) {
                            struct stat st;
                            int fd;
                            if (mode == 1) {
                                fstatat(dfd, n.c_str(), &st, AT_SYMLINK_NOFOLLOW);
                                fd = openat(dfd, n.c_str(), O_RDONLY | O_CLOEXEC);
                            } else {
                                fd = openat(dfd, n.c_str(), O_RDONLY | O_CLOEXEC | O_NOFOLLOW | O_NONBLOCK);
                                fstat(fd, &st);
                            }
                            bytes += pread(fd, buf.data(), (size_t)st.st_size, 0);
                            close(fd);
                        }
                        closedir(dir);
                    }
                });
            for (auto& x : th) x.join();
            const double dt = now() - t0;
            printf("mode %d, %2d threads%s: %6.1f ms for %d files = %.2f us per file (wall), %.2f us per file and thread, %.1f GB/s\n", mode, nt,
                   mode == 3 ? (unshared == nt ? " (own fd tables)" : " (unshare REFUSED)") : "", dt * 1e3, nd * per, dt / (nd * per) * 1e6,
                   dt / (nd * per) * 1e6 * nt, bytes / dt / 1e9);
            fflush(stdout);
        }
    const std::string rm = "rm -rf " + root;
    return system(rm.c_str());
}
