// tools/power_sampler.hpp -- development aid (round 4): socket power and shader clock of the GPU while a measurement runs.
// Reads the amdgpu hwmon files of the GPU with the given PCI address (power1_input or power1_average in microwatts, freq1_input =
// sclk in Hz, power1_cap) from a sampler thread every `period_ms`; `amd-smi metric -p -c` reports the same numbers.
#pragma once
#include <ctype.h>
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

struct PowerStats {
    int n = 0;
    double mean_w = 0, max_w = 0, min_w = 0, mean_sclk_mhz = 0, min_sclk_mhz = 0, max_sclk_mhz = 0, cap_w = 0;
};

class PowerSampler {
    std::string power_path_, freq_path_, cap_path_;
    std::thread th_;
    std::atomic<bool> stop_{false};
    std::vector<double> w_, f_;
    int period_ms_;

    static bool read_ll(const std::string& p, long long* v)
    {
        FILE* f = fopen(p.c_str(), "r");
        if (!f) return false;
        const int ok = fscanf(f, "%lld", v);
        fclose(f);
        return ok == 1;
    }

public:
    // `pci_bdf`: "0000:d9:00.0" of the device the measurement runs on (hipDeviceGetPCIBusId): a box shows the hwmon
    // directories of every GPU of the host, and the first card's is somebody else's GPU
    explicit PowerSampler(const char* pci_bdf, int period_ms = 10) : period_ms_(period_ms)
    {
        std::string bdf = pci_bdf ? pci_bdf : "";
        for (auto& ch : bdf) ch = (char)tolower(ch);
        for (int pass = 0; pass < 1 && !bdf.empty(); pass++) {
            const std::string base = "/sys/bus/pci/devices/" + bdf + "/hwmon";
            DIR* d = opendir(base.c_str());
            if (!d) continue;
            while (dirent* e = readdir(d)) {
                if (strncmp(e->d_name, "hwmon", 5) != 0) continue;
                const std::string h = base + "/" + e->d_name;
                long long v;
                for (const char* nm : {"/power1_input", "/power1_average"})
                    if (power_path_.empty() && read_ll(h + nm, &v)) power_path_ = h + nm;
                if (!power_path_.empty()) {
                    freq_path_ = h + "/freq1_input";
                    cap_path_ = h + "/power1_cap";
                    break;
                }
            }
            closedir(d);
        }
    }
    bool available() const { return !power_path_.empty(); }
    void start()
    {
        w_.clear();
        f_.clear();
        stop_ = false;
        if (!available()) return;
        th_ = std::thread([this] {
            while (!stop_) {
                long long p = 0, f = 0;
                if (read_ll(power_path_, &p)) w_.push_back(p * 1e-6);
                if (read_ll(freq_path_, &f)) f_.push_back(f * 1e-6);
                std::this_thread::sleep_for(std::chrono::milliseconds(period_ms_));
            }
        });
    }
    // `skip_frac`: leading share of the samples dropped (the ramp from idle)
    PowerStats finish(double skip_frac = 0.25)
    {
        PowerStats s;
        if (!available()) return s;
        stop_ = true;
        th_.join();
        long long cap = 0;
        if (read_ll(cap_path_, &cap)) s.cap_w = cap * 1e-6;
        const size_t k0 = (size_t)(w_.size() * skip_frac);
        for (size_t i = k0; i < w_.size(); i++) {
            s.mean_w += w_[i];
            if (s.n == 0 || w_[i] > s.max_w) s.max_w = w_[i];
            if (s.n == 0 || w_[i] < s.min_w) s.min_w = w_[i];
            s.n++;
        }
        if (s.n) s.mean_w /= s.n;
        const size_t f0 = (size_t)(f_.size() * skip_frac);
        int nf = 0;
        for (size_t i = f0; i < f_.size(); i++) {
            s.mean_sclk_mhz += f_[i];
            if (nf == 0 || f_[i] > s.max_sclk_mhz) s.max_sclk_mhz = f_[i];
            if (nf == 0 || f_[i] < s.min_sclk_mhz) s.min_sclk_mhz = f_[i];
            nf++;
        }
        if (nf) s.mean_sclk_mhz /= nf;
        return s;
    }
};
