function [uk,lk] = warmup_class2(c,r,l,p,q,mu,phi,res,maxit)
% Drop-in shim (Class2/warmup_class2.m:1; nargin rules of :3-17 resolved here).
if nargin == 7, res = 1e-1; maxit = inf; end
if nargin == 8, maxit = inf; end
if res == 0 && maxit == inf, error('res = 0 and maxit = inf'); end
ipd_mex('apd_create', 2, c, r, l, double(p), double(q), mu, double(phi));
[uk,lk] = ipd_mex('apd_warmup', res, maxit);
end
