function [xk,lk] = warmup_class1(c,r,l,p,q,gama,res,maxit)
% Drop-in shim (Class1/warmup_class1.m:2; nargin rules of :3-20 resolved here).
if nargin == 6, res = 1e-1; maxit = inf; end
if nargin == 7, maxit = inf; end
if res == 0 && maxit == inf, error('res = 0 and maxit = inf'); end
ipd_mex('apd_create', 1, c, r, l, double(p), double(q), gama);
[xk,lk] = ipd_mex('apd_warmup', res, maxit);
end
