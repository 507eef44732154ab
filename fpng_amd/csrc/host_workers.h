// host_workers.h -- the encoder's copy threads: from pageable memory an "asynchronous" copy keeps its caller busy for most of its
// duration, so the host paths (pipeline.cpp: fpng_amd_encode_host_to; decode_api.cpp: the uploads of fpng_amd_decode_batch and the
// streamed fpng_amd_decode_host) issue their copies from threads of their own, made once per encoder and kept.
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

// A thread that runs one task at a time (fpng_amd_encode_host_to's uploader / downloader)
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> task;
    bool busy = false, quit = false;
    void loop()
    {
        for (;;) {
            std::function<void()> t;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return busy || quit; });
                if (!busy) return;
                t = std::move(task);
            }
            t();
            {
                std::lock_guard<std::mutex> lk(mu);
                busy = false;
            }
            cv.notify_all();
        }
    }
    void start(std::function<void()> t)
    {
        if (!th.joinable()) th = std::thread([this] { loop(); });
        {
            std::lock_guard<std::mutex> lk(mu);
            task = std::move(t);
            busy = true;
        }
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !busy; });
    }
    ~Worker()
    {
        if (!th.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_all();
        th.join();
    }
};
struct HostWorkers {
    Worker up, down;
};

